"""The N>1 path on CPU: world_size-2 gloo processes shard a batch the way the reference's
tower loop does (main.py:511-517) and all-gather the per-rank coords (main.py:573-574)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import joint_cnn_mrf_amd  # noqa: F401
from joint_cnn_mrf_amd import dist as jdist
from oracle import jcm_oracle as O


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, batch, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        # every rank builds the same global batch of heat maps, keeps its slice, takes the
        # argmax with the oracle (stand-in for the GPU forward), gathers coords
        hm = np.random.RandomState(3).standard_normal((batch, 60, 90, 9)).astype(np.float32)
        mine = jdist.shard_batch(torch.as_tensor(hm))
        lo, hi = jdist.shard_bounds(batch, world, rank)
        assert mine.shape[0] == hi - lo == batch // world
        local = torch.as_tensor(O.argmax_coords(mine.numpy()))
        allc = jdist.allgather_coords(local)
        q.put((rank, allc.numpy()))
    finally:
        dist.destroy_process_group()


def test_shard_and_allgather_world2():
    world, batch = 2, 6
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, batch, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    hm = np.random.RandomState(3).standard_normal((batch, 60, 90, 9)).astype(np.float32)
    ref = O.argmax_coords(hm)
    for r in range(world):
        assert res[r].shape == (batch, 2, 9) and res[r].dtype == np.int32
        np.testing.assert_array_equal(res[r], ref)      # rank-major order == tf.concat(axis=0)


def test_shard_bounds_drop_remainder_like_reference():
    # imgs_per_gpu = batch_size // n_gpus (main.py:511): 14 images on 4 towers -> 3 each, 2 dropped
    assert [jdist.shard_bounds(14, 4, r) for r in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 12)]
    assert jdist.shard_bounds(2048, 8, 7) == (1792, 2048)


def test_allgather_is_identity_without_process_group():
    c = torch.zeros(3, 2, 9, dtype=torch.int32)
    assert jdist.allgather_coords(c) is c


# ---------------------------------------------------------------- training: tower-averaged gradients
def _train_case():
    from joint_cnn_mrf_amd import synth
    p = synth.make_pd_params(debug=True, bn='trained')
    rs = np.random.RandomState(21)
    x = rs.random_sample((4, 32, 48, 3)).astype(np.float32)
    y = np.zeros((4, 4, 6, 10), np.float32)
    for b in range(4):
        for k in range(10):
            y[b, rs.randint(4), rs.randint(6), k] = 1.0
    return p, x, y


def _flat(grads):
    return np.concatenate([np.asarray(grads[k], np.float64).reshape(-1) for k in sorted(grads)])


def _train_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from oracle import train_oracle as T
        p, x, y = _train_case()
        lo, hi = jdist.shard_bounds(x.shape[0], world, rank)
        r = T.loss_and_grads(x[lo:hi], y[lo:hi], p, use_sm=False)       # the tower's compute_gradients (stand-in for the GPU)
        g = torch.as_tensor(_flat(r['grads']))
        jdist.average_gradients(g)
        q.put((rank, g.numpy()))
    finally:
        dist.destroy_process_group()


def test_average_gradients_world2_equals_two_towers():
    """One process per GPU + all-reduce/N is the reference's in-graph tower average (main.py:243-267)."""
    from oracle import train_oracle as T
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_train_worker, args=(r, world, port, q)) for r in range(world)]
    for p_ in procs:
        p_.start()
    res = dict(q.get(timeout=300) for _ in range(world))
    for p_ in procs:
        p_.join(timeout=60)
        assert p_.exitcode == 0
    p, x, y = _train_case()
    _, info = T.train_step(x, y, p, {}, 1, use_sm=False, towers=2)
    want = _flat(info['grads'])
    for r in range(world):
        np.testing.assert_allclose(res[r], want, rtol=1e-12, atol=1e-15)


def test_average_gradients_is_identity_without_process_group():
    g = torch.arange(5, dtype=torch.float32)
    assert jdist.average_gradients(g) is g
