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


class RcclComm:
    """A libjcm RCCL communicator (include/jcm.h: jcm_comm_*): the C-ABI route of the coordinate all-gather, for hosts that
    do not want torch.distributed on the data path.  The 128-byte rendezvous id travels through the torch.distributed
    store (any backend) once; afterwards `allgather_coords` is a single ncclAllGather on the engine's stream."""

    def __init__(self, engine, group=None):
        import ctypes
        from . import _lib
        self._lib, self._eng = _lib.load(), engine
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        ident = ctypes.create_string_buffer(128)
        if rank == 0:
            _lib.check(self._lib.jcm_comm_unique_id(ident), 'jcm_comm_unique_id')
        if world > 1:
            box = [ident.raw]
            dist.broadcast_object_list(box, src=0, group=group)
            ident = ctypes.create_string_buffer(box[0], 128)
        h = ctypes.c_void_p()
        _lib.check(self._lib.jcm_comm_create(ident, world, rank, engine.device.index, ctypes.byref(h)), 'jcm_comm_create')
        self._h, self.world, self.rank = h, world, rank

    def allgather_coords(self, local_coords):
        """[B_local,2,K] int32 on the engine's device -> [world*B_local,2,K], rank-major (main.py:573-574)."""
        import ctypes
        from . import _lib
        if local_coords.dtype != torch.int32 or local_coords.device != self._eng.device or not local_coords.is_contiguous():
            raise TypeError('coords must be a contiguous int32 tensor on %s' % self._eng.device)
        out = torch.empty((self.world * local_coords.shape[0],) + tuple(local_coords.shape[1:]), dtype=torch.int32, device=local_coords.device)
        _lib.check(self._lib.jcm_allgather_coords(self._eng._h, self._h, ctypes.c_void_p(local_coords.data_ptr()), local_coords.shape[0],
                                                  ctypes.c_void_p(out.data_ptr())), 'jcm_allgather_coords')
        return out

    def close(self):
        if getattr(self, '_h', None) is not None and self._h.value:
            self._lib.jcm_comm_destroy(self._h)
            self._h.value = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Towers:
    """The reference's in-process towers (main.py:509-517,573-574): `--gpus i j k` builds one tower per listed device inside
    ONE process, slices the batch into batch_size // n_gpus contiguous pieces and concatenates the per-tower results.
    Here a tower is an Engine (its own replica of the parameters) on its device; the launches of all towers are enqueued
    before any result is read, so the devices run concurrently.  (A device may be listed twice: two towers share it.)"""

    def __init__(self, params, gpus, **engine_kw):
        from .engine import Engine
        if not gpus:
            raise ValueError('--gpus needs at least one device index')
        self.gpus = [int(g) for g in gpus]
        self.engines = [Engine(device=g, **engine_kw).load_params(params) for g in self.gpus]

    @property
    def n(self):
        return len(self.engines)

    def slices(self, batch_size):
        return [shard_bounds(batch_size, self.n, i) for i in range(self.n)]           # main.py:511,516-517

    def forward(self, x, torso, use_sm=True, want_prob=False):
        """x [B,480,720,3], torso [B,60,90,1] (host or any device) -> dict of tensors on the FIRST tower's device,
        concatenated in tower order (= tf.concat axis 0); a remainder B % n_gpus is dropped as in the reference."""
        outs = []
        for eng, (lo, hi) in zip(self.engines, self.slices(x.shape[0])):
            xs = torch.as_tensor(x[lo:hi]).to(eng.device, non_blocking=True).contiguous()
            ts = torch.as_tensor(torso[lo:hi]).to(eng.device, non_blocking=True).contiguous() if use_sm else None
            with torch.cuda.device(eng.device):
                outs.append(eng.forward(xs, ts, use_sm=use_sm, want_prob=want_prob))
        dev0 = self.engines[0].device
        return {k: torch.cat([o[k].to(dev0) for o in outs], dim=0) for k in outs[0]}

    def close(self):
        for e in self.engines:
            e.close()
