"""The callers either side of the hot path, on the GPU: eval_error (main.py:275-283), the in-process towers of
`--gpus i j ...` (main.py:509-517,573-574) for inference and for the training step, the RCCL communicator of the C ABI,
and the command line end to end (train on generated data, save a tf.train.Saver checkpoint, restore it, evaluate)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import joint_cnn_mrf_amd  # noqa: F401
from joint_cnn_mrf_amd import synth
from oracle import jcm_oracle as O
from oracle import train_oracle as T

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _params(kind='trained'):
    p = synth.make_pd_params(debug=True, bn='trained', conv6_gain=8.0)
    p.update(synth.make_sm_params(synth.synthetic_priors(), kind=kind))
    return p


@pytest.mark.parametrize('use_sm', [True, False])
def test_eval_error_matches_oracle(use_sm):
    """Means over whole batches of (loss_pd, loss_sm, det_rate_pd, det_rate_sm), remainder dropped (5 examples, batch 2)."""
    from joint_cnn_mrf_amd import evaluation
    from joint_cnn_mrf_amd.engine import Engine
    p = _params()
    X, Y = synth.make_images(5, seed=41), synth.make_targets(5, seed=42)
    eng = Engine(device=0).load_params(p)
    got = evaluation.eval_error(X, Y, eng, 2, use_sm=use_sm, joints=[2], det_radius=10)
    got_all = evaluation.eval_error(X, Y, eng, 2, use_sm=use_sm, joints='all', det_radius=25)
    eng.close()
    want, want_all = np.zeros(4), np.zeros(4)
    for lo in (0, 2):                                       # the fifth example is dropped (get_next_batch, main.py:184-192)
        x, y = X[lo:lo + 2].astype(np.float64), Y[lo:lo + 2].astype(np.float64)
        r = O.forward(x, y[..., 9:], p, use_sm=use_sm)
        nchw = lambda a: torch.as_tensor(np.ascontiguousarray(a.transpose(0, 3, 1, 2)))
        l_pd = float(T.softmax_cross_entropy(nchw(r['pd_logits']), nchw(y[..., :9])))
        l_sm = float(T.softmax_cross_entropy(nchw(r['sm_logits']), nchw(y[..., :9]))) if use_sm else l_pd
        sm_prob = r['sm_prob'] if use_sm else r['pd_prob']
        want += [l_pd, l_sm, O.det_rate(r['pd_prob'], y[..., :9], 10, [2]), O.det_rate(sm_prob, y[..., :9], 10, [2])]
        want_all += [l_pd, l_sm, O.det_rate(r['pd_prob'], y[..., :9], 25, 'all'), O.det_rate(sm_prob, y[..., :9], 25, 'all')]
    np.testing.assert_allclose(got[:2], want[:2] / 2, rtol=2e-5)
    np.testing.assert_allclose(got[2:], want[2:] / 2, atol=1e-4)
    np.testing.assert_allclose(got_all, want_all / 2, rtol=2e-5, atol=1e-4)


def test_in_process_towers_equal_one_tower():
    """`--gpus 0 0`: two towers (both on the only device of the box), slices of batch_size // 2, tf.concat order; a remainder
    image is dropped like the reference's static slices."""
    from joint_cnn_mrf_amd.dist import Towers
    from joint_cnn_mrf_amd.engine import Engine
    p = _params()
    x, torso = synth.make_images(5, seed=51), synth.make_torso(5, seed=52)
    tw = Towers(p, [0, 0])
    r = tw.forward(x, torso, use_sm=True, want_prob=True)
    tw.close()
    eng = Engine(device=0).load_params(p)
    one = eng.forward(torch.as_tensor(x[:4], device='cuda:0'), torch.as_tensor(torso[:4], device='cuda:0'), use_sm=True)
    eng.close()
    assert r['sm_coords'].shape == (4, 2, 9)
    for k in ('pd_coords', 'sm_coords', 'pd_prob'):
        assert torch.equal(r[k], one[k]), k
    assert float((r['sm_prob'] - one['sm_prob']).abs().max()) <= 1e-6


def test_tower_training_step_matches_oracle_two_towers():
    """One update of two in-process towers against the restated step with towers=2 (gradients averaged, moving statistics
    advanced tower after tower, main.py:243-267,557)."""
    from joint_cnn_mrf_amd.dist import Towers
    from joint_cnn_mrf_amd.main import TowerTrainer
    p = _params()
    x, y = synth.make_images(2, seed=61), synth.make_targets(2, seed=62)      # one image per tower (the float64 restatement is what this test spends its time on)
    want, info = T.train_step(x, y, p, {}, 1, lr=0.001, lmbd=0.001, use_sm=True, optimizer='adam', towers=2)
    tw = Towers(p, [0, 0])
    tt = TowerTrainer(tw, p, optimizer='adam', lr=0.001, lmbd=0.001, use_sm=True)
    tt.train_step(x, y)
    got = [tr.get_params(p) for tr in tt.trainers]
    avg = tt.trainers[0].grads_dict()                  # the tower-averaged gradients the update used
    tw.close()
    shapes = {k: np.asarray(v).shape for k, v in p.items()}
    for k in want:
        np.testing.assert_array_equal(got[0][k], got[1][k], err_msg=k)          # the replicas stay identical
    # average_gradients: exactly the mean of what single towers compute on the two halves (same kernels, same order of
    # operations), and the restated towers=2 gradient up to the fp32-vs-float64 ReLU / max-pool decisions that
    # test_gpu_train.py accounts for per tensor
    from joint_cnn_mrf_amd.engine import Engine
    from joint_cnn_mrf_amd.train import Trainer
    halves = []
    for lo in (0, 1):
        e1 = Engine(device=0).load_params(p)
        t1 = Trainer(e1, optimizer='adam', lr=0.001, lmbd=0.001, use_sm=True)
        t1.loss_and_grads(torch.as_tensor(x[lo:lo + 1], device='cuda:0'), torch.as_tensor(y[lo:lo + 1], device='cuda:0'))
        halves.append(t1.grads.clone())
        e1.close()
    mean = halves[0].clone()
    mean += halves[1]
    mean /= 2
    flat = np.concatenate([avg[n] for n, _o, _c in tt.trainers[0].layout])
    np.testing.assert_array_equal(flat, mean.cpu().numpy())
    for k, g in info['grads'].items():
        g = np.asarray(g, np.float64).reshape(-1)
        assert np.abs(avg[k] - g).max() <= 2e-2 * np.abs(g).max() + 1e-7, k
    for k in want:                                                              # the towers' update ops, tower 0 first
        if k.endswith('moving_mean') or k.endswith('moving_variance'):
            np.testing.assert_allclose(got[0][k], np.asarray(want[k]).reshape(shapes[k]), rtol=2e-5, atol=1e-6, err_msg=k)
    # clip + Adam on exactly those averaged gradients (Adam's first step moves every weight by ~lr whatever the size of
    # its gradient, so the update is checked from the gradients the GPU produced, as in test_apply_gradients)
    cur = {k: np.asarray(v, np.float64) for k, v in p.items()}
    clipped, _norm = T.clip_by_global_norm({k: v.astype(np.float64).reshape(shapes[k]) for k, v in avg.items()})
    upd = T.adam_apply(cur, clipped, {}, 1, 0.001)
    for k, v in upd.items():
        np.testing.assert_allclose(got[0][k], v, rtol=2e-6, atol=2e-7, err_msg=k)


def test_rccl_communicator_c_abi_one_rank():
    """jcm_comm_* / jcm_allgather_coords (include/jcm.h): RCCL initialises through the C ABI and the all-gather runs on the
    engine's stream (one rank: the box has one GPU and RCCL refuses two ranks on one device)."""
    from joint_cnn_mrf_amd.dist import RcclComm
    from joint_cnn_mrf_amd.engine import Engine
    eng = Engine(device=0)
    eng.finalize()
    comm = RcclComm(eng)
    c = torch.randint(0, 60, (7, 2, 9), dtype=torch.int32, device='cuda:0')
    out = comm.allgather_coords(c)
    assert comm.world == 1 and out.shape == (7, 2, 9) and torch.equal(out, c) and out.data_ptr() != c.data_ptr()
    comm.close()
    eng.close()


def _cli(args, cwd):
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, '-m', 'joint_cnn_mrf_amd.main'] + args, cwd=cwd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout


def test_cli_train_save_restore_evaluate(tmp_path):
    """`--train --synthetic --debug` for two epochs on two towers: the reference's progress line per epoch, a
    tf.train.Saver checkpoint after epoch 2 (> n_epochs // 2); `--restore` resumes from it with n_iters carried over; an
    evaluation run writes predictions.mat.  `--data_augm` and unknown devices are refused, not ignored."""
    from joint_cnn_mrf_amd import tf_checkpoint
    common = ['--debug', '--use_sm', '--synthetic', '--synthetic_size', '8', '--batch_size', '4', '--model_path', str(tmp_path / 'models_ex')]
    out = _cli(['--train', '--gpus', '0', '0', '--n_epochs', '2'] + common, str(tmp_path))
    lines = [l for l in out.splitlines() if l.startswith('Epoch ')]
    assert [l.split()[1] for l in lines] == ['0', '1', '2'] and 'test_dr' in lines[0] and 'train_mse' in lines[0]
    ckpts = sorted(f for f in os.listdir(tmp_path / 'models_ex') if f.endswith('.index'))
    assert len(ckpts) == 1 and ckpts[0].endswith('-2.index')
    prefix = str(tmp_path / 'models_ex' / ckpts[0][:-len('.index')])
    state = tf_checkpoint.load_checkpoint(prefix)
    assert int(state['n_iters']) == 2 * (8 // 4) and 'conv5/weights/Adam_1' in state and 'energy_lsho_lelb' in state
    out2 = _cli(['--train', '--restore', '--restore_path', prefix, '--gpus', '0', '--n_epochs', '1'] + common, str(tmp_path))
    assert len([l for l in out2.splitlines() if l.startswith('Epoch ')]) == 2
    pred = str(tmp_path / 'matlab' / 'predictions.mat')
    out3 = _cli(['--restore', '--restore_path', prefix, '--gpus', '0', '0', '--predictions', pred] + common, str(tmp_path))
    rec = json.loads(out3.strip().splitlines()[-1])
    import scipy.io
    m = scipy.io.loadmat(pred)
    assert m['flic_pred_pd'].shape == (2, 9, rec['n_images']) and m['flic_pred_sm'].shape == (2, 9, rec['n_images'])
    env = dict(os.environ, PYTHONPATH=ROOT)
    for bad in (['--train', '--data_augm', '--gpus', '0'] + common, ['--gpus', '99'] + common):
        r = subprocess.run([sys.executable, '-m', 'joint_cnn_mrf_amd.main'] + bad, cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode != 0
