"""Joint training step on the GPU against oracle/train_oracle.py
(float64 autograd restatement of main.py:511-577).  Gradients are compared per tensor,
relative to that tensor's largest entry: 1e-4, the heat-map tolerance of the forward path."""
import numpy as np
import pytest
import torch

import joint_cnn_mrf_amd  # noqa: F401
from golden_util import sampled_conv_grads
from joint_cnn_mrf_amd import synth
from oracle import train_oracle as T

pytestmark = pytest.mark.gpu

GRAD_RTOL = 1e-4


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a), device='cuda:0')


def make_trainer(params, f32_conv=None, precision='fp32', conv9_fft=None, **kw):
    from joint_cnn_mrf_amd.engine import Engine
    from joint_cnn_mrf_amd.train import Trainer
    eng = Engine(device=0, precision=precision, f32_conv=f32_conv, split_min_wgs=0 if f32_conv else None, conv9_fft=conv9_fft).load_params(params)
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


@pytest.fixture(scope='module')
def joint_ref(debug_case):
    """The float64 restatement of the joint step at --debug size (shared: it costs ~25 s of CPU)."""
    p, x, y = debug_case
    return T.loss_and_grads(x, y, p, use_sm=True, lmbd=0.001)


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


def test_joint_loss_and_grads(debug_case, joint_ref):
    """use_sm: loss_sm flows through the spatial model into the 81 priors / biases, bn_sm and, through
    hm_pred_pd, back into the part detector (main.py:523-531,539)."""
    p, x, y = debug_case
    ref = joint_ref
    ref32 = T.loss_and_grads(x, y, p, use_sm=True, lmbd=0.001, dtype=torch.float32)
    eng, tr = make_trainer(p, use_sm=True, lmbd=0.001)
    losses, _ = tr.loss_and_grads(dev(x), dev(y))
    got = tr.grads_dict()
    l = losses.cpu().numpy()
    eng.close()
    np.testing.assert_allclose(l, [ref['loss'], ref['loss_pd'], ref['loss_sm'], ref['l2']], rtol=2e-5)
    check_grads(got, ref['grads'], ref32['grads'], verbose=True)


def test_moving_statistics_update(debug_case, joint_ref):
    """UPDATE_OPS (main.py:557): moving = 0.9*moving + 0.1*batch, variance Bessel-corrected."""
    p, x, y = debug_case
    ref = joint_ref
    want = T.update_moving(p, ref['bn_stats'])
    eng, tr = make_trainer(p, use_sm=True)
    tr.loss_and_grads(dev(x), dev(y))
    got = {k: tr.get_tensor(k, np.asarray(p[k]).shape) for k in want}
    eng.close()
    for k in want:
        np.testing.assert_allclose(got[k], want[k], rtol=2e-5, atol=1e-7, err_msg=k)


@pytest.mark.parametrize('optimizer', ['adam', 'momentum'])
def test_apply_gradients(debug_case, optimizer):
    """grad_renorm(4.0) + apply_gradients (main.py:576-577) on the gradients the GPU produced: the
    restated tf.train update applied to the same numbers must land on the same parameters, two updates
    in a row (slots carried over).  Then every derived table must have followed: the inference tower
    evaluated with the new parameters equals the oracle's."""
    import oracle.jcm_oracle as O
    p, x, y = debug_case
    eng, tr = make_trainer(p, use_sm=True, optimizer=optimizer, lr=0.001)
    cur = {k: np.asarray(v, np.float64) for k, v in p.items()}
    slots = {}
    shapes = {k: np.asarray(v).shape for k, v in p.items()}
    for step in (1, 2):
        tr.loss_and_grads(dev(x), dev(y))
        g = {k: v.astype(np.float64).reshape(shapes[k]) for k, v in tr.grads_dict().items()}
        for k in list(cur):                       # moving statistics were advanced by the forward
            if k.endswith('moving_mean') or k.endswith('moving_variance'):
                cur[k] = tr.get_tensor(k, shapes[k]).astype(np.float64)
        clipped, norm = T.clip_by_global_norm(g)
        upd = T.adam_apply(cur, clipped, slots, step, 0.001) if optimizer == 'adam' else T.momentum_apply(cur, clipped, slots, 0.001)
        cur.update(upd)
        got_norm = tr.apply(want_norm=True)
        assert abs(got_norm - norm) <= 1e-5 * norm
        assert tr.n_iters == step
        got = tr.get_params(p)
        for k in cur:
            np.testing.assert_allclose(got[k], cur[k], rtol=2e-6, atol=2e-7, err_msg='%s after update %d' % (k, step))
    xs, torso = x[:1], y[:1, :, :, 9:]
    r = eng.forward(dev(xs), dev(torso), use_sm=True)
    ref = O.forward(xs.astype(np.float64), torso.astype(np.float64), {k: v.astype(np.float64) for k, v in got.items()})
    eng.close()
    np.testing.assert_allclose(r['pd_prob'].cpu().numpy(), ref['pd_prob'], atol=1e-4, rtol=0)
    np.testing.assert_allclose(r['sm_prob'].cpu().numpy(), ref['sm_prob'], atol=1e-4, rtol=0)


def test_loss_goes_down(debug_case):
    p, x, y = debug_case
    eng, tr = make_trainer(p, use_sm=True, optimizer='adam', lr=0.001)
    xs, ys = dev(x), dev(y)
    hist = []
    for _ in range(6):
        losses, _ = tr.train_step(xs, ys)
        hist.append(losses.cpu().numpy().copy())
    eng.close()
    hist = np.array(hist)
    assert np.isfinite(hist).all()
    assert hist[-1, 0] < hist[0, 0], hist[:, 0]


def test_det_rate_matches_restatement():
    """evaluation.py:4-37 through the arg-max kernel."""
    import oracle.jcm_oracle as O
    from joint_cnn_mrf_amd import evaluation
    from joint_cnn_mrf_amd.engine import Engine
    rs = np.random.RandomState(8)
    tgt = synth.make_targets(16)[..., :9].copy()
    pred = tgt + 0.02 * rs.random_sample(tgt.shape).astype(np.float32)       # mostly right, some joints displaced
    pred[::3] = rs.random_sample(pred[::3].shape).astype(np.float32)
    eng = Engine(device=0)
    eng.finalize()
    for joints in ('all', [2], [2, 5, 8]):
        for radius in (10, 30):
            got = evaluation.det_rate(dev(pred), dev(tgt), radius, joints, engine=eng)
            assert abs(got - O.det_rate(pred, tgt, radius, joints)) < 1e-4
    eng.close()


@pytest.mark.parametrize('f32_conv', ['exact', 'chain', 'split16'])
def test_full_size_step_vs_golden(f32_conv):
    """The full-size network (filters 64..512), one 480x720 image: losses, sampled gradient entries, gradient
    norms and the moving-statistics update against tests/golden/train_full.json (float64 restatement,
    generated by tests/golden/make_train_golden.py).  'exact' = the default fp32 engine: forward, data gradient AND weight gradient of the
    stride-1 layers in the frequency domain (conv_fft.hip, wgrad_fft.hip; two scaled fp16 parts per operand); 'split16' = the direct
    fp16x3 kernels; 'chain' = everything on the fp32 MFMA chain (conv9_fft off), whose rounding the golden's float32 slack was measured with and
    which is therefore held to the strict bound.  The other routes get 1e-2 HERE ONLY, for the end-to-end comparison of ONE image, where a
    single ReLU / max-pool decision that rounds the other way moves a gradient of the 15x23 maps; the gradient KERNELS themselves are held to
    1e-6 in test_gradient_kernels_at_full_size_layer_shapes and the 16-image step to the strict bound in test_full_size_step_16_images_vs_golden."""
    import json, os
    from golden.make_train_golden import case, LMBD
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'train_full.json')) as fh:
        gold = json.load(fh)
    p, x, y = case()
    eng, tr = make_trainer(p, f32_conv='exact' if f32_conv == 'chain' else f32_conv, conv9_fft=False if f32_conv == 'chain' else None, use_sm=True,
                           lmbd=LMBD)      # 'split16': forward + data gradient of conv4_fullres / conv5 on conv_split.hip
    assert eng.conv_kernel_name('conv5', 1, 60, 90).startswith('conv_fft') == (f32_conv == 'exact')
    losses, _ = tr.loss_and_grads(dev(x), dev(y))
    got = tr.grads_dict()
    l = losses.cpu().numpy()
    moving = {k: tr.get_tensor(k, np.asarray(p[k]).shape).reshape(-1) for k in gold['moving']}
    eng.close()
    np.testing.assert_allclose(l, gold['losses'], rtol=2e-5)
    assert set(gold['tensors']) == set(got)
    bad = []
    for k, t in gold['tensors'].items():
        g = got[k].astype(np.float64)
        # 'chain': the strict bound.  The frequency-domain route and the split modes compute the same convolutions to fp32 accuracy (pinned
        # layer by layer in test_conv_layer_random_shape / test_split_kernels_match_exact_on_every_layer_shape, and for every gradient at
        # debug size in test_joint_loss_and_grads) but round differently, so a ReLU / max-pool decision can flip
        # somewhere else than it does in torch's fp32 run; on the 15x23 maps of one image (345 samples per channel) one flip
        # moves a gradient by up to ~1e-2 of its largest entry (measured 8e-3 split, 3e-3 frequency domain: the quarter-resolution
        # branch only, identical with the direct and the frequency-domain data gradient -- it is the forward's rounding).
        # (3 x slack since round 5: the batch statistics are summed in double now, so the fp32 run whose distance to float64 `slack` measures is no
        # longer reproduced decision for decision; three beta gradients sat at 2.2-2.6 x slack)
        tol = (GRAD_RTOL if f32_conv == 'chain' else 1e-2) * t['max'] + 3 * t['slack'] + 1e-7
        err = np.abs(g[t['idx']] - np.asarray(t['val'])).max()
        nerr = abs(np.linalg.norm(g) - t['norm'])
        if not (err <= tol and nerr <= tol * np.sqrt(g.size)):
            bad.append('%s: entry err %.3e norm err %.3e (tol %.3e, max|g| %.3e)' % (k, err, nerr, tol, t['max']))
    assert not bad, '\n'.join(bad)
    for k, t in gold['moving'].items():
        np.testing.assert_allclose(moving[k][t['idx']], t['val'], rtol=2e-5, atol=1e-7, err_msg=k)


def test_full_size_step_bf16_mixed_precision():
    """bf16 handle: bf16 activations / gradients between the layers, bf16 MFMA with fp32 accumulate; fp32 master weights,
    BatchNorm statistics, losses, spatial model and optimizer.  Against the float64 golden: losses to 1 %; per gradient
    tensor the cosine similarity over up to 512 sampled entries and the norm.  Bias gradients of layers followed by
    BatchNorm are pure cancellation noise (the batch mean removes the bias: their true gradient is ~0) and are only
    required to stay small."""
    import json, os
    from golden.make_train_golden import case, LMBD
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    with open(os.path.join(here, 'train_full.json')) as fh:
        gold = json.load(fh)
    dense = np.load(os.path.join(here, 'train_full_samples.npz'))
    p, x, y = case()
    eng, tr = make_trainer(p, precision='bf16', use_sm=True, lmbd=LMBD)
    losses, _ = tr.loss_and_grads(dev(x), dev(y))
    got = tr.grads_dict()
    l = losses.cpu().numpy()
    eng.close()
    np.testing.assert_allclose(l, gold['losses'], rtol=1e-2)
    rows, bad = [], []
    for k, t in gold['tensors'].items():
        g = got[k].astype(np.float64)
        a, b = g[dense[k + '|idx']], dense[k + '|val'].astype(np.float64)
        cos = float(a @ b / max(np.linalg.norm(a) * np.linalg.norm(b), 1e-300))
        nerr = abs(np.linalg.norm(g) - t['norm']) / max(t['norm'], 1e-30)
        rows.append((cos, nerr, k))
        if k.endswith('/biases') and k != 'conv6/biases':
            ok = np.linalg.norm(g) <= 3 * t['norm'] + 1e-6
        elif k == 'conv6/biases':
            ok = np.abs(g).max() < 1e-5                      # exactly zero in exact arithmetic (softmax is shift invariant)
        else:
            ok = cos >= 0.96 and nerr <= 0.10        # measured: worst cosine 0.975, worst norm error 0.092
        if not ok:
            bad.append('%s: cosine %.4f, norm err %.3f (norm %.2e)' % (k, cos, nerr, t['norm']))
    for r in sorted(rows)[:10]:
        print('  cosine %.4f  norm err %.3f  %s' % r)
    assert not bad, '\n'.join(bad)


@pytest.mark.parametrize('mode', ['exact', 'split16', 'bf16'])
def test_gradients_are_deterministic_and_training_is_stable(mode):
    """The same batch twice gives bit-identical gradients (fixed-order split-K reductions, no atomics), and 12 Adam
    updates at full size keep every loss finite and moving down in all three arithmetic modes."""
    p = synth.make_pd_params(debug=False, bn='identity')
    p.update(synth.make_sm_params(synth.synthetic_priors(), kind='init'))
    kw = dict(precision='bf16') if mode == 'bf16' else dict(f32_conv=mode)
    eng, tr = make_trainer(p, use_sm=True, lr=0.001, **kw)
    x, y = dev(synth.make_images(2, seed=5)), dev(synth.make_targets(2, seed=6))
    tr.loss_and_grads(x, y)
    g1 = tr.grads.clone()
    tr.loss_and_grads(x, y)
    assert torch.equal(g1, tr.grads)
    hist = []
    for _ in range(12):
        losses, norm = tr.train_step(x, y, want_norm=True)
        hist.append(losses.cpu().numpy().copy())
        assert np.isfinite(norm)
    eng.close()
    hist = np.array(hist)
    assert np.isfinite(hist).all()
    assert hist[-1, 1] < hist[0, 1] and hist[-1, 2] < hist[0, 2], hist[:, :3]


@pytest.mark.parametrize('use_sm', [True, False])
def test_gradient_ready_notifications_cover_every_tensor_once(debug_case, use_sm):
    """jcm_train_set_grad_callback: the ranges reported during the backward pass (the hook the RCCL overlap hangs on) are
    disjoint, aligned with whole tensors, reported last-layer-first, and together cover every trainable element that the
    loss reaches -- and at the moment a range is reported its kernels are already enqueued (the value read after a
    stream sync inside the hook equals the final gradient)."""
    p, x, y = debug_case
    eng, tr = make_trainer(p, use_sm=use_sm)
    seen, snaps = [], []

    def hook(off, cnt):
        seen.append((off, cnt))
        torch.cuda.synchronize()
        snaps.append(tr.grads[off:off + cnt].clone())

    tr.set_ready_hook(hook)
    tr.loss_and_grads(dev(x), dev(y))
    torch.cuda.synchronize()
    final = tr.grads.clone()
    eng.close()
    starts = {o: (n, c) for n, o, c in tr.layout}
    covered = np.zeros(tr.n_elements, bool)
    for (off, cnt), snap in zip(seen, snaps):
        assert off in starts and not covered[off:off + cnt].any()
        covered[off:off + cnt] = True
        assert torch.equal(snap, final[off:off + cnt])
    names_left = [n for n, o, c in tr.layout if not covered[o]]
    if use_sm:
        assert not names_left
        assert tr.layout[[o for _, o, _ in tr.layout].index(seen[0][0])][0].startswith('bias_')      # spatial model first
    else:
        assert names_left and all(n.startswith(('bias_', 'bn_sm', 'energy_')) for n in names_left)
    order = [starts[o][0].split('/')[0] for o, _ in seen if not starts[o][0].startswith(('bias_', 'bn_sm', 'energy_'))]
    assert order[0] == 'conv6' and order[1] == 'conv5' and order[-1] == 'conv1_quarterres'


def test_border_clipped_targets_follow_tf_gradient(debug_case):
    """Targets whose blob is cut by the map border do not sum to one (data.py:171-183).  TF-1.x's
    softmax_cross_entropy_with_logits back-propagates softmax - labels there, NOT the exact derivative
    softmax * sum(labels) - labels; the reference trains with TF's, and so must the kernels (the two differ by the
    missing mass: 7/16 for a corner blob)."""
    p, x, y = debug_case
    y2 = y.copy()
    kern = np.outer([1, 2, 1], [1, 2, 1]).astype(np.float32) / 16
    y2[:, :, :, 0] = 0
    y2[:, 0:2, 0:2, 0] = kern[1:, 1:]            # corner: 9/16 of the mass left
    y2[:, :, :, 4] = 0
    y2[:, 58:60, 40:43, 4] = kern[:2, :]         # bottom edge: 12/16
    ref = T.loss_and_grads(x, y2, p, use_sm=True, lmbd=0.001)
    ref32 = T.loss_and_grads(x, y2, p, use_sm=True, lmbd=0.001, dtype=torch.float32)
    eng, tr = make_trainer(p, use_sm=True, lmbd=0.001)
    losses, _ = tr.loss_and_grads(dev(x), dev(y2))
    got = tr.grads_dict()
    eng.close()
    np.testing.assert_allclose(losses.cpu().numpy()[:3], [ref['loss'], ref['loss_pd'], ref['loss_sm']], rtol=2e-5)
    check_grads(got, ref['grads'], ref32['grads'])
    # part detector alone (loss = 2 * CE_pd, main.py:535): d loss / d conv6 bias_k = 2/(B*K) * sum_b (sum_px p - sum_px t)
    # = 2/K * (1 - mass_k) with TF's gradient; the exact derivative p * sum(t) - t would give 0 for every joint.
    eng, tr = make_trainer(p, use_sm=False, lmbd=0.0)
    tr.loss_and_grads(dev(x), dev(y2))
    gb = tr.grads_dict()['conv6/biases']
    eng.close()
    want = np.zeros(9)
    want[0], want[4] = 2.0 / 9 * (1 - 9.0 / 16), 2.0 / 9 * (1 - 12.0 / 16)
    np.testing.assert_allclose(gb, want, atol=2e-6)


def test_optimizer_state_round_trip_resumes_training(debug_case):
    """Saver.save / Saver.restore cover the Adam slots, the beta powers and n_iters (main.py:604,612,666): three
    uninterrupted steps equal two steps, a save, a fresh session restored from it, and the third step -- bit for bit."""
    from joint_cnn_mrf_amd import checkpoint
    p, x, y = debug_case
    xd, yd = dev(x), dev(y)
    eng, tr = make_trainer(p, use_sm=True, lmbd=0.001, n_updates_total=3)       # 3 updates: the LR schedule moves at every one
    for _ in range(3):
        tr.train_step(xd, yd)
    want = tr.get_params(p)
    eng.close()
    eng, tr = make_trainer(p, use_sm=True, lmbd=0.001, n_updates_total=3)
    for _ in range(2):
        tr.train_step(xd, yd)
    saved = checkpoint.session_state(tr, p)
    assert saved['n_iters'] == 2 and 'conv5/weights/Adam' in saved and 'conv5/weights/Adam_1' in saved and 'beta1_power' in saved
    eng.close()
    params2 = {k: v for k, v in saved.items() if k in p}
    eng, tr = make_trainer(params2, use_sm=True, lmbd=0.001, n_updates_total=3)
    checkpoint.restore_session_state(tr, saved)
    assert tr.n_iters == 2
    tr.train_step(xd, yd)
    got = tr.get_params(p)
    eng.close()
    for k in want:
        np.testing.assert_array_equal(got[k], want[k], err_msg=k)


# ---- the gradient KERNELS alone, at full-size layer shapes and the per-GPU batch of configs[4] (16 images)
LAYER_SHAPES = [('conv5', 60, 90), ('conv4_fullres', 60, 90), ('conv6', 60, 90), ('conv3_fullres', 60, 90), ('conv2_fullres', 120, 180),
                ('conv4_halfres', 30, 45), ('conv2_halfres', 60, 90), ('conv4_quarterres', 15, 23), ('conv3_quarterres', 15, 23)]


@pytest.mark.parametrize('mode', ['exact', 'exact_nowin', 'chain', 'split16'])
def test_gradient_kernels_at_full_size_layer_shapes(mode):
    """jcm_train_layer_grads: the weight-gradient and data-gradient kernels of every stride-1 layer shape of the full-width network on the SAME
    x and dz, 16 images (a tower's share of configs[4]: the batch is the K axis of the frequency-domain weight gradient's per-frequency product,
    and a training handle scales all 16 images of a tensor by one power of two) -- compared at sampled entries with float64 sums taken straight
    from the definition.  No ReLU or pooling is involved, so there is no rounding-decision noise: the frequency-domain kernels are held to 1e-6 of
    the tensor's largest entry, the direct kernels to 8e-6 (100x / 12x below GRAD_RTOL).  'exact' = the default fp32 engine (frequency
    domain: wgrad_fft.hip and the data gradient through conv_fft on flipped filters, two scaled fp16 parts), 'chain' = wgrad.hip / conv_igemm on the fp32 MFMA accumulation chain, 'split16' = the direct kernels on fp16 parts.  The default engine runs
    the wide 60x90 layers (conv4_fullres, conv5) on 32x32 overlap-save windows (jcm_train.hip: 544 frequencies, 192 window "images"); 'exact_nowin'
    (fft_windows = 0) keeps them on the 64x96 transform of the whole map, as round 3 did -- both are held to the same 1e-6."""
    p = synth.make_pd_params(debug=False, bn='trained')
    p.update(synth.make_sm_params(synth.synthetic_priors(), kind='init'))
    lmbd = 0.001
    kw = dict(f32_conv='exact' if mode in ('exact', 'exact_nowin', 'chain') else mode, conv9_fft=False if mode == 'chain' else None)
    eng, tr = make_trainer(p, use_sm=True, lmbd=lmbd, **kw)
    if mode == 'exact_nowin':
        eng.set_option('fft_windows', 0)
    B = 16
    g = torch.Generator(device='cuda:0')
    rows = []
    for li, (scope, H, W) in enumerate(LAYER_SHAPES):
        w = np.asarray(p[scope + '/weights'])
        k, _, cin, cout = w.shape
        assert eng.conv_kernel_name(scope, B, H, W).startswith('conv_fft') == (mode in ('exact', 'exact_nowin')), scope
        g.manual_seed(100 + li)
        x = torch.relu(torch.randn((B, H, W, cin), device='cuda:0', generator=g))
        # dz as BatchNorm's backward leaves it: zero mean per channel (the weight gradient is then a sum with heavy cancellation), scale ~1e-3
        dz = torch.randn((B, H, W, cout), device='cuda:0', generator=g) * 1e-3
        dz = (dz - dz.mean(dim=(0, 1, 2), keepdim=True)).contiguous()
        dw, dx = tr.layer_grads(scope, x, dz)
        dx = dx.cpu().numpy()
        (wi, wv), (xi, xv) = sampled_conv_grads(x.cpu().numpy(), dz.cpu().numpy(), w, lmbd, np.random.RandomState(li))
        ew = np.abs(dw[wi].astype(np.float64) - wv).max() / np.abs(dw).max()
        ex = np.abs(np.array([dx[i] for i in xi], np.float64) - xv).max() / np.abs(dx).max()
        rows.append((scope, ew, ex))
        print('  %-7s %-18s dW err / max|dW| %.2e   dX err / max|dX| %.2e' % (mode, scope, ew, ex))
    eng.close()
    # measured (round 4, 16 images): frequency domain dW <= 2.6e-7, dX <= 2.6e-7 (both operand forms); fp32 MFMA chain dW <= 9.3e-7, dX <= 2.1e-6;
    # direct fp16x3 kernels dW <= 1.0e-6, dX <= 2.3e-6 -- the bound is ~4x the measured value of each route, far below GRAD_RTOL
    bound = 1e-6 if mode in ('exact', 'exact_nowin') else 8e-6
    bad = [r for r in rows if not (r[1] <= bound and r[2] <= bound)]
    assert not bad, (bound, bad)


def test_full_size_step_16_images_vs_golden():
    """configs[4] at its real per-GPU workload (main.py:538-541,557-560: batch 128 over 8 towers): the full-width network, 16 images, training
    handle, default route -- losses, moving statistics, gradient norms and sampled gradient entries against tests/golden/train_full_b16.json
    (float64 restatement of the 16-image tower; training-mode BatchNorm couples the images, so this golden is its own two-hour CPU run).
    End-to-end bound, BOTH routes: GRAD_RTOL = 1e-4 of the tensor's largest entry + three times the float32 restatement's own distance from float64
    (i.e. ~5e-3 of max|g| end to end -- the gradient KERNELS are held to 1e-6 in test_gradient_kernels_at_full_size_layer_shapes) (at 16
    images x 5400 samples per channel a single rounding-decision flip no longer moves a gradient the way it does in the one-image golden; measured
    worst entry: frequency domain 5.4e-4 of max|g| with a float32 slack of 2.6e-3, fp32 MFMA chain 3.8e-3 with a slack of 3.0e-3)."""
    import json, os
    from golden.make_train_golden import case, LMBD, B_TOWER
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'train_full_b16.json')
    with open(path) as fh:
        gold = json.load(fh)
    p, x, y = case(B_TOWER)
    for mode in ('exact', 'chain'):
        eng, tr = make_trainer(p, conv9_fft=False if mode == 'chain' else None, use_sm=True, lmbd=LMBD)
        losses, _ = tr.loss_and_grads(dev(x), dev(y))
        got = tr.grads_dict()
        l = losses.cpu().numpy()
        moving = {k: tr.get_tensor(k, np.asarray(p[k]).shape).reshape(-1) for k in gold['moving']}
        eng.close()
        np.testing.assert_allclose(l, gold['losses'], rtol=2e-5)
        bad, rows = [], []
        for k, t in gold['tensors'].items():
            g = got[k].astype(np.float64)
            tol = GRAD_RTOL * t['max'] + 3 * t['slack'] + 1e-7      # (3 x: see test_full_size_step_vs_golden)
            err = np.abs(g[t['idx']] - np.asarray(t['val'])).max()
            nerr = abs(np.linalg.norm(g) - t['norm'])
            rows.append((err / max(t['max'], 1e-30), k, t['slack'] / max(t['max'], 1e-30)))
            if not (err <= tol and nerr <= tol * np.sqrt(g.size)):
                bad.append('%s %s: entry err %.3e norm err %.3e (tol %.3e, max|g| %.3e)' % (mode, k, err, nerr, tol, t['max']))
        for r in sorted(rows, reverse=True)[:6]:
            print('  %s rel %.2e  %-40s f32-slack %.2e' % ((mode,) + r))
        assert not bad, '\n'.join(bad)
        for k, t in gold['moving'].items():
            np.testing.assert_allclose(moving[k][t['idx']], t['val'], rtol=2e-5, atol=1e-7, err_msg=k)


def test_gradient_ready_callback_may_call_the_library(debug_case):
    """The gradient-ready callback (jcm_train_set_grad_callback) runs WITHOUT the per-device call lock (round 4 held it through the whole entry point,
    so a callback that called jcm_get_tensor dead-locked): a hook that reads a parameter through the C ABI and uses a second engine inside the
    callback returns, every trainable element is reported exactly once, and the gradients equal those of a step without a hook."""
    from joint_cnn_mrf_amd.engine import Engine
    p, x, y = debug_case
    eng, tr = make_trainer(p, use_sm=True, lmbd=0.001)
    tr.loss_and_grads(dev(x), dev(y))
    want = tr.grads.clone()
    other = Engine(device=0).load_params(p)
    seen, reads = [], []

    def hook(offset, count):
        seen.append((offset, count))
        reads.append(float(tr.get_tensor('conv6/biases', (9,)).sum()))                 # same handle, read-only entry point
        other.spatial_softmax(torch.zeros((1, 60, 90, 9), device='cuda:0'))            # another handle's entry point on the same device
    tr.set_ready_hook(hook)
    tr.loss_and_grads(dev(x), dev(y))
    tr.set_ready_hook(None)
    other.close()
    assert torch.equal(tr.grads, want)
    cover = np.zeros(tr.n_elements, np.int32)
    for o, c in seen:
        cover[o:o + c] += 1
    assert (cover == 1).all() and len(reads) == len(seen) > 10
    eng.close()


def test_gradient_ready_callback_cannot_reenter_the_running_step(debug_case):
    """A callback that calls an entry point of the SAME handle which uses the workspace arena or the training state (forward, conv_layer, another
    step, an update) would overwrite the running step's activations: such calls return JCM_ERR_STATE (RuntimeError in the binding) instead of
    corrupting the step (round 5's advisor finding), read-only entry points stay allowed, and the step's gradients are those of a step without a hook."""
    p, x, y = debug_case
    eng, tr = make_trainer(p, use_sm=True, lmbd=0.001)
    tr.loss_and_grads(dev(x), dev(y))
    want = tr.grads.clone()
    errors, calls = [], [0]

    def hook(offset, count):
        calls[0] += 1
        if calls[0] != 3:
            return
        for fn in (lambda: eng.model(dev(x)),
                   lambda: eng.spatial_model(torch.zeros((1, 60, 90, 10), device='cuda:0')),
                   lambda: tr.loss_and_grads(dev(x), dev(y)),
                   lambda: eng.update_tensor('conv6/biases', np.zeros(9, np.float32))):
            try:
                fn()
                errors.append(None)
            except RuntimeError as e:
                errors.append(str(e))
        float(tr.get_tensor('conv6/biases', (9,)).sum())      # read-only: allowed
    tr.set_ready_hook(hook)
    tr.loss_and_grads(dev(x), dev(y))
    tr.set_ready_hook(None)
    assert len(errors) == 4 and all(e is not None and 'gradient-ready callback' in e for e in errors), errors
    assert torch.equal(tr.grads, want)
    eng.close()


def test_two_host_threads_on_one_handle_are_serialised():
    """A handle is documented as not thread-safe, but two host threads that call it were serialised by the per-call lock in rounds 1-4; round 5's
    nesting check (an unsynchronised depth counter) misread the second thread as a nested call.  Nesting is now decided by the calling THREAD and the
    handle carries its own mutex for the duration of an outermost call: forwards from two threads on one engine give the single-thread results."""
    import threading
    from joint_cnn_mrf_amd.engine import Engine
    p = synth.make_pd_params(debug=True, bn='trained', conv6_gain=8.0)
    eng = Engine(device=0).load_params(p)
    xs = [dev(synth.make_images(2, seed=70 + i)) for i in range(2)]
    want = [eng.model(xi).clone() for xi in xs]
    got, errs = [[None] * 6, [None] * 6], []

    def work(t):
        try:
            torch.cuda.set_device(0)
            for i in range(6):
                got[t][i] = eng.model(xs[t]).clone()
            torch.cuda.synchronize()
        except Exception as e:      # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for t in range(2):
        for i in range(6):
            assert torch.equal(got[t][i], want[t]), (t, i)
    eng.close()


def test_window_route_follows_the_batch_and_other_map_sizes():
    """jcm_train.hip: takes_windows() sends a wide layer to 32x32 overlap-save windows for B <= 32 and to the whole-map transform above; the other
    geometry's filter spectra stay in the cache (bounded by JCM_FFT_CACHE_GB) when the route flips.  conv5's weight and data gradient through jcm_train_layer_grads: B = 16 (windows),
    B = 33 (whole map), B = 16 again on ONE handle (both geometries' spectra stay cached: round 6) -- the first and the third result bit-identical, the B = 33 result equal to a fresh handle's and
    within 1e-6 of float64 sums at sampled entries; then a 48x72 map (2 x 3 windows) with and without windows against the same float64 sums."""
    p = synth.make_pd_params(debug=False, bn='trained')
    p.update(synth.make_sm_params(synth.synthetic_priors(), kind='init'))
    lmbd = 0.001
    w = np.asarray(p['conv5/weights'])
    g = torch.Generator(device='cuda:0')

    def tensors(B, H, W, seed):
        g.manual_seed(seed)
        x = torch.relu(torch.randn((B, H, W, 512), device='cuda:0', generator=g))
        dz = torch.randn((B, H, W, 512), device='cuda:0', generator=g) * 1e-3
        return x, (dz - dz.mean(dim=(0, 1, 2), keepdim=True)).contiguous()

    def check(dw, dx, x, dz, seed):
        (wi, wv), (xi, xv) = sampled_conv_grads(x.cpu().numpy(), dz.cpu().numpy(), w, lmbd, np.random.RandomState(seed), n=24)
        dxn = dx.cpu().numpy()
        ew = np.abs(dw[wi].astype(np.float64) - wv).max() / np.abs(dw).max()
        ex = np.abs(np.array([dxn[i] for i in xi], np.float64) - xv).max() / np.abs(dxn).max()
        assert ew <= 1e-6 and ex <= 1e-6, (ew, ex)

    eng, tr = make_trainer(p, use_sm=True, lmbd=lmbd)
    x16, z16 = tensors(16, 60, 90, 1)
    x33, z33 = tensors(33, 60, 90, 2)
    dw_a, dx_a = tr.layer_grads('conv5', x16, z16)
    dx_a = dx_a.clone()
    dw_b, dx_b = tr.layer_grads('conv5', x33, z33)
    dx_b = dx_b.clone()
    dw_c, dx_c = tr.layer_grads('conv5', x16, z16)
    assert np.array_equal(dw_a, dw_c) and torch.equal(dx_a, dx_c)
    check(dw_b, dx_b, x33, z33, 5)
    xs, zs = tensors(4, 48, 72, 3)
    dw_w, dx_w = tr.layer_grads('conv5', xs, zs)
    check(dw_w, dx_w, xs, zs, 6)
    eng.set_option('fft_windows', 0)
    dw_n, dx_n = tr.layer_grads('conv5', xs, zs)
    check(dw_n, dx_n, xs, zs, 6)
    eng.close()
    eng2, tr2 = make_trainer(p, use_sm=True, lmbd=lmbd)
    dw_f, dx_f = tr2.layer_grads('conv5', x33, z33)
    assert np.array_equal(dw_b, dw_f) and torch.equal(dx_b, dx_f)
    eng2.close()
