"""Batch sharding across GPUs: the inference side of the reference's tower loop.

The reference slices the batch into `batch_size // n_gpus` contiguous pieces, one per
in-graph tower, and concatenates the per-tower heat maps on axis 0 (main.py:511-517,573-574).
Images are independent in inference (BatchNorm uses moving statistics, main.py:406), so
here every rank (one process per GPU) runs the whole path on its slice with no data-path
collective, and only the argmax coordinates -- [B_local,2,K] int32, 72 B/image -- are
all-gathered (RCCL over xGMI via torch.distributed 'nccl'; 'gloo' in the CPU tests)."""
import torch
import torch.distributed as dist


def shard_bounds(batch_size, world_size, rank):
    """main.py:511,516: imgs_per_gpu = batch_size // n_gpus; id_from, id_to = i*per, i*per+per.
    (Like the reference, a remainder batch_size % n_gpus is dropped.)"""
    per = batch_size // world_size
    return rank * per, rank * per + per


def shard_batch(x, world_size=None, rank=None):
    """The rank's contiguous slice of a global batch tensor (first axis)."""
    world_size = dist.get_world_size() if world_size is None else world_size
    rank = dist.get_rank() if rank is None else rank
    lo, hi = shard_bounds(x.shape[0], world_size, rank)
    return x[lo:hi]


def _on_stream(stream):
    """Collectives are enqueued on torch's CURRENT stream; the tensors they move were produced on the engine's
    stream.  Making that stream current orders the collective behind the kernels (no event needed)."""
    import contextlib
    return torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()


def allgather_coords(local_coords, group=None, stream=None):
    """[B_local,2,K] int32 per rank -> [world*B_local,2,K] on every rank, rank-major order,
    i.e. the order tf.concat(hms_pred, axis=0) gives (main.py:573-574).  `stream`: the torch stream the
    coordinates were produced on (Engine._stream) when it is not the current one.  With an initialised
    process group the collective runs for any world size, 1 included (the same code path on every box)."""
    if not dist.is_available() or not dist.is_initialized():
        return local_coords
    if local_coords.dtype != torch.int32:
        raise TypeError('coords must be int32')
    world = dist.get_world_size(group)
    local_coords = local_coords.contiguous()
    dev = local_coords.device
    if dist.get_backend(group) != 'nccl' and dev.type != 'cpu':
        if stream is not None:
            stream.synchronize()
        local_coords = local_coords.cpu()            # gloo moves host memory (CPU tests / plumbing runs)
        stream = None
    out = torch.empty((world * local_coords.shape[0],) + tuple(local_coords.shape[1:]), dtype=local_coords.dtype,
                      device=local_coords.device)
    with _on_stream(stream):
        dist.all_gather_into_tensor(out, local_coords, group=group)
    return out.to(dev)


def average_gradients(flat_grads, group=None, stream=None):
    """average_gradients (main.py:243-267) across ranks: the mean of the per-tower gradients, in place
    on the flat fp32 gradient buffer -- one all-reduce (RCCL over xGMI: 58.7 M fp32 = 235 MB) per step.
    The training step is the one place on this path with a real exchange; BatchNorm statistics stay
    per tower, as in the reference."""
    if not dist.is_available() or not dist.is_initialized():
        return flat_grads
    world = dist.get_world_size(group)
    with _on_stream(stream if flat_grads.device.type != 'cpu' else None):      # `stream`: where the gradients were produced
        if dist.get_backend(group) != 'nccl' and flat_grads.device.type != 'cpu':
            host = flat_grads.cpu()                   # gloo moves host memory (CPU tests / plumbing runs)
            dist.all_reduce(host, group=group)
            flat_grads.copy_(host)
        else:
            dist.all_reduce(flat_grads, group=group)
        if world > 1:
            flat_grads.div_(world)
    return flat_grads
