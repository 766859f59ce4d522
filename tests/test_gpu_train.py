"""Joint training step on the GPU against oracle/train_oracle.py
(float64 autograd restatement of main.py:511-577).  Gradients are compared per tensor,
relative to that tensor's largest entry: 1e-4, the heat-map tolerance of the forward path."""
import numpy as np
import pytest
import torch

import joint_cnn_mrf_amd  # noqa: F401
from joint_cnn_mrf_amd import synth
from oracle import train_oracle as T

pytestmark = pytest.mark.gpu

GRAD_RTOL = 1e-4


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a), device='cuda:0')


def make_trainer(params, **kw):
    from joint_cnn_mrf_amd.engine import Engine
    from joint_cnn_mrf_amd.train import Trainer
    eng = Engine(device=0).load_params(params)
    return eng, Trainer(eng, **kw)


def check_grads(got, ref64, ref32=None, rtol=GRAD_RTOL, atol=1e-7, verbose=False):
    """Per tensor: |got - ref64| <= rtol * max|ref64| + atol (+ 2 * the float32 restatement's own distance
    from float64 when `ref32` is given: a ReLU / max-pool decision that flips between float32 and float64
    moves a gradient by more than rounding, and the reference computes in float32)."""
    bad, rows = [], []
    for k, g in ref64.items():
        a = got[k].reshape(-1).astype(np.float64)
        b = np.asarray(g, np.float64).reshape(-1)
        scale = np.abs(b).max()
        err = np.abs(a - b).max()
        slack = 0.0 if ref32 is None else 2 * np.abs(np.asarray(ref32[k], np.float64).reshape(-1) - b).max()
        rows.append((err / max(scale, 1e-30), k, scale, slack / max(scale, 1e-30)))
        if not err <= rtol * scale + atol + slack:
            bad.append('%s: err %.3e (max |g| %.3e, f32 slack %.3e)' % (k, err, scale, slack))
    if verbose or bad:
        for r in sorted(rows, reverse=True)[:12]:
            print('  rel %.2e  %-40s max|g| %.3e  f32-slack %.2e' % r)
    assert not bad, '\n'.join(bad)


@pytest.fixture(scope='module')
def debug_case():
    p = synth.make_pd_params(debug=True, bn='trained')
    p.update(synth.make_sm_params(synth.synthetic_priors(), kind='trained'))
    B = 2
    return p, synth.make_images(B), synth.make_targets(B)


def test_pd_only_loss_and_grads(debug_case):
    p, x, y = debug_case
    ref = T.loss_and_grads(x, y, p, use_sm=False, lmbd=0.001)
    ref32 = T.loss_and_grads(x, y, p, use_sm=False, lmbd=0.001, dtype=torch.float32)
    eng, tr = make_trainer(p, use_sm=False, lmbd=0.001)
    losses, _ = tr.loss_and_grads(dev(x), dev(y))
    got = tr.grads_dict()
    l = losses.cpu().numpy()
    eng.close()
    np.testing.assert_allclose(l, [ref['loss'], ref['loss_pd'], ref['loss_sm'], ref['l2']], rtol=2e-5)
    pd_only = lambda d: {k: v for k, v in d.items() if not (k.startswith('energy_') or k.startswith('bias_') or k.startswith('bn_sm'))}
    check_grads(got, pd_only(ref['grads']), pd_only(ref32['grads']), verbose=True)
    for k in got:                                    # the loss does not reach the spatial model: exact zeros
        if k.startswith('energy_') or k.startswith('bias_') or k.startswith('bn_sm'):
            assert not got[k].any(), k


def test_joint_loss_and_grads(debug_case):
    """use_sm: loss_sm flows through the spatial model into the 81 priors / biases, bn_sm and, through
    hm_pred_pd, back into the part detector (main.py:523-531,539)."""
    p, x, y = debug_case
    ref = T.loss_and_grads(x, y, p, use_sm=True, lmbd=0.001)
    ref32 = T.loss_and_grads(x, y, p, use_sm=True, lmbd=0.001, dtype=torch.float32)
    eng, tr = make_trainer(p, use_sm=True, lmbd=0.001)
    losses, _ = tr.loss_and_grads(dev(x), dev(y))
    got = tr.grads_dict()
    l = losses.cpu().numpy()
    eng.close()
    np.testing.assert_allclose(l, [ref['loss'], ref['loss_pd'], ref['loss_sm'], ref['l2']], rtol=2e-5)
    check_grads(got, ref['grads'], ref32['grads'], verbose=True)
