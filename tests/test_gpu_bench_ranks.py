"""The command the driver runs for the scaling curve -- `bench.py --gpus N` with N > 1 -- executed on the one GPU of the box: bench.py starts its
own `torch.distributed.run` ranks (main.py:509-517,573-574 is the reference's tower loop), JCM_BENCH_BACKEND=gloo lets both ranks share cuda:0
(coordinates cross through host memory; the real multi-GPU run is nccl = RCCL).  Checked: exactly ONE JSON line on stdout (rank 0's, the last
line, <= 4 KB), n_gpus / global batch / scaling as asked, finite values, and nothing printed by rank 1."""
import json
import math
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*flags, timeout=900):
    env = dict(os.environ, JCM_BENCH_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0', MASTER_ADDR='127.0.0.1')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + list(flags), cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, 'bench.py %s failed (%d)\n%s\n%s' % (' '.join(flags), r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    json_lines = [ln for ln in lines if ln.lstrip().startswith('{')]
    assert len(json_lines) == 1 and lines[-1] == json_lines[0], 'want exactly one JSON line, last on stdout; got %d:\n%s' % (len(json_lines), r.stdout[-2000:])
    assert len(json_lines[0]) <= 4096, len(json_lines[0])
    return json.loads(json_lines[0]), r


def _finite(v):
    return isinstance(v, (int, float)) and math.isfinite(v) and v > 0


def test_bench_two_ranks_weak_scaling_line():
    d, _ = _bench('--gpus', '2', '--dtype', 'bf16', '--batch', '4', '--steps', '2', '--warmup', '1', '--cpu-reps', '0')
    assert d['n_gpus'] == 2 and d['scaling'] == 'weak' and d['steps'] == 2 and d['warmup'] == 1
    assert d['config']['batch_per_gpu'] == 4 and d['config']['global_batch'] == 8
    assert _finite(d['value']) and _finite(d['ms_per_step']) and _finite(d['median_ms_per_step'])
    assert abs(d['value'] - 8 / (d['ms_per_step'] * 1e-3)) <= 1e-3 * d['value']            # whole-job images / max-over-ranks time
    assert d['roofline']['bound'] in ('hbm', 'mfma') and _finite(d['roofline']['frac'])
    assert 'cpu_baseline' not in d                                                            # an N = 1 item


def test_bench_two_ranks_fixed_global_batch():
    d, _ = _bench('--gpus', '2', '--global-batch', '8', '--steps', '2', '--warmup', '1', '--cpu-reps', '0')
    assert d['n_gpus'] == 2 and d['scaling'] == 'strong'
    assert d['config']['batch_per_gpu'] == 4 and d['config']['global_batch'] == 8
    assert 'configs[3]' in d['config']['workload'] and _finite(d['value'])


def test_bench_two_ranks_training_step():
    d, _ = _bench('--train', '--gpus', '2', '--batch', '2', '--debug', '--steps', '2', '--warmup', '1')
    assert d['n_gpus'] == 2 and d['config']['batch_per_gpu'] == 2 and d['config']['global_batch'] == 4
    assert 'training' in d['metric'] and _finite(d['value']) and _finite(d['ms_per_step'])
    assert 'all_reduce' in d['config']['collective']


def test_bench_one_rank_nccl_branch():
    """The `nccl` branch of bench.py, as far as one GPU allows: `torch.distributed.run --nproc-per-node 1` with
    JCM_BENCH_FORCE_DIST=1 makes bench.py create a ONE-rank RCCL group -- init_process_group('nccl', device_id=cuda:0), the barriers around the timed
    regions, the GPU-side all_reduce(MAX) of the step time, the all-gather of the coordinates over RCCL and destroy_process_group all execute, on the
    fixed-global-batch line (configs[3] form) and on the training step's gradient all-reduce."""
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, JCM_BENCH_FORCE_DIST='1', HSA_ENABLE_IPC_MODE_LEGACY='0', MASTER_ADDR='127.0.0.1')
    env.pop('JCM_BENCH_BACKEND', None)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    base = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1', '--master-port', str(port),
            os.path.join(ROOT, 'bench.py'), '--gpus', '1']
    for flags in (['--global-batch', '8', '--steps', '2', '--warmup', '1', '--cpu-reps', '0'],
                  ['--train', '--batch', '2', '--debug', '--steps', '2', '--warmup', '1']):
        r = subprocess.run(base + flags, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, 'bench.py %s failed (%d)\n%s\n%s' % (' '.join(flags), r.returncode, r.stdout[-2000:], r.stderr[-4000:])
        json_lines = [ln for ln in r.stdout.splitlines() if ln.lstrip().startswith('{')]
        assert len(json_lines) == 1, r.stdout[-2000:]
        d = json.loads(json_lines[0])
        assert d['n_gpus'] == 1 and _finite(d['value']) and _finite(d['ms_per_step'])
