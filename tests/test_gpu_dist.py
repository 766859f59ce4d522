"""The multi-rank path ON the GPU box (one MI355X): two ranks that share cuda:0 run the real
Engine.forward on their tower slices (main.py:511-517) and all-gather the coordinates (main.py:573-574);
a one-rank RCCL group exercises the 'nccl' branches -- RCCL initialises and every collective of the path
(coordinate all-gather, gradient all-reduce, the overlapped per-layer all-reduce of the training step)
executes on the device.  No scaling curve can be measured on one GPU; this pins correctness only."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import joint_cnn_mrf_amd  # noqa: F401

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _case(batch):
    from joint_cnn_mrf_amd import synth
    p = synth.make_pd_params(debug=True, bn='trained', conv6_gain=8.0)
    p.update(synth.make_sm_params(synth.synthetic_priors(), kind='trained'))
    return p, synth.make_images(batch, seed=31), synth.make_torso(batch, seed=32)


def _run(target, world, *args):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, q) + args) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


def _shard_worker(rank, world, port, q, batch, backend):
    import torch.distributed as dist
    from joint_cnn_mrf_amd import dist as jdist
    from joint_cnn_mrf_amd.engine import Engine
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    if backend == 'nccl':
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', 0))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        p, x, torso = _case(batch)
        eng = Engine(device=0).load_params(p)
        xg, tg = torch.as_tensor(x, device='cuda:0'), torch.as_tensor(torso, device='cuda:0')
        mine = eng.forward(jdist.shard_batch(xg).contiguous(), jdist.shard_batch(tg).contiguous(), use_sm=True, want_prob=False)
        allc = jdist.allgather_coords(mine['sm_coords'])
        whole = eng.forward(xg, tg, use_sm=True, want_prob=False)['sm_coords'] if rank == 0 else None
        torch.cuda.synchronize()
        q.put((rank, (allc.cpu().numpy(), None if whole is None else whole.cpu().numpy(), str(allc.device))))
        eng.close()
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_one_gpu_forward_shards_and_allgather():
    """world_size 2 (gloo; both ranks on cuda:0): Engine.forward on shard_batch slices + allgather_coords equals
    the single-rank forward of the whole batch, in tf.concat order -- on every rank."""
    batch = 6
    res = _run(_shard_worker, 2, batch, 'gloo')
    whole = res[0][1]
    assert whole.shape == (batch, 2, 9)
    for r in range(2):
        got, _w, device = res[r]
        assert got.dtype == np.int32 and device.startswith('cuda')
        np.testing.assert_array_equal(got, whole)
    # and the single-rank result is the oracle's
    from oracle import jcm_oracle as O
    p, x, torso = _case(batch)
    np.testing.assert_array_equal(whole, O.forward(x, torso, p)['sm_coords'])


def test_rccl_one_rank_group_allgather():
    """backend 'nccl' = RCCL: the branch of allgather_coords the 8-GPU run takes, on a one-rank group."""
    batch = 3
    res = _run(_shard_worker, 1, batch, 'nccl')
    got, whole, device = res[0]
    assert device.startswith('cuda')
    np.testing.assert_array_equal(got, whole)


def _train_worker(rank, world, port, q):
    import torch.distributed as dist
    from joint_cnn_mrf_amd import synth
    from joint_cnn_mrf_amd.engine import Engine
    from joint_cnn_mrf_amd.train import Trainer
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', 0))
    try:
        p = synth.make_pd_params(debug=True, bn='trained')
        p.update(synth.make_sm_params(synth.synthetic_priors(), kind='trained'))
        x = torch.as_tensor(synth.make_images(2), device='cuda:0')
        y = torch.as_tensor(synth.make_targets(2), device='cuda:0')
        moving = Trainer.moving_statistics_of(p)
        out = {}
        for overlap in (False, True):
            eng = Engine(device=0).load_params(p)
            tr = Trainer(eng, optimizer='adam', lr=1e-3, lmbd=1e-3, use_sm=True, overlap_allreduce=overlap)
            ranges = []
            tr.set_ready_hook(lambda off, cnt: ranges.append((off, cnt)))
            for _ in range(2):
                del ranges[:]
                losses, norm = tr.train_step(x, y, want_norm=True, moving=moving)
            torch.cuda.synchronize()
            out[overlap] = (tr.get_params(p), losses.cpu().numpy(), norm, len(tr._pending), sum(c for _o, c in ranges), tr.n_elements)
            eng.close()
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_rccl_overlapped_gradient_allreduce_equals_plain_step():
    """Trainer(overlap_allreduce=True) end to end on RCCL (one-rank group): per-layer all-reduces started from the
    gradient-ready notifications on a side stream, waited for before the update -- parameters, losses and the
    gradient norm after two steps equal the non-overlapped step's exactly."""
    res = _run(_train_worker, 1)[0]
    (p0, l0, n0, pend0, cov0, ne0), (p1, l1, n1, pend1, cov1, ne1) = res[False], res[True]
    assert cov0 == ne0 and cov1 == ne1          # every trainable element reported exactly once per pass
    assert pend0 == 0 and pend1 == 0            # nothing left in flight after the step
    np.testing.assert_array_equal(l0, l1)
    assert n0 == n1
    for k in p0:
        np.testing.assert_array_equal(p0[k], p1[k], err_msg=k)
