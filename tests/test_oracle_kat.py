"""Closed-form known-answer tests that pin the oracle's TF-1.x semantics (SURVEY.md 8c
KAT1-KAT8), plus agreement of the two independently written formulations."""
import numpy as np
import pytest
import torch

import joint_cnn_mrf_amd  # noqa: F401  (registers the package alias)
from joint_cnn_mrf_amd import synth
from oracle import jcm_oracle as O
from oracle import jcm_oracle_torch as T


def test_kat1_conv_mrf_delta_shifts_prior():
    """B = delta(u0,v0)  =>  Cpre[y,x] = A[y+59-u0, x+89-v0]  (main.py:83-87)."""
    rs = np.random.RandomState(0)
    A = rs.random_sample((1, 120, 180, 1))
    for (u0, v0) in [(0, 0), (59, 89), (17, 42)]:
        Bm = np.zeros((1, 60, 90, 1))
        Bm[0, u0, v0, 0] = 1.0
        pre = O.conv_mrf_pre(A, Bm)[0, :, :, 0]
        assert pre.shape == (61, 91)
        np.testing.assert_array_equal(pre, A[0, 59 - u0:120 - u0, 89 - v0:180 - v0, 0])


def test_kat1b_conv_mrf_is_true_convolution():
    from scipy.signal import convolve2d
    rs = np.random.RandomState(1)
    A = rs.random_sample((1, 120, 180, 1))
    Bm = rs.random_sample((2, 60, 90, 1))
    pre = O.conv_mrf_pre(A, Bm)
    for b in range(2):
        ref = convolve2d(A[0, :, :, 0], Bm[b, :, :, 0], mode='valid')
        np.testing.assert_allclose(pre[b, :, :, 0], ref, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(O.conv_mrf(A, Bm), T.conv_mrf(A, Bm), rtol=1e-12, atol=1e-12)


def test_kat2_resize_tables():
    """61->60: src = i*61/60; 30->60: even rows copy, odd rows average, last clamps;
    2x down-sampling is pure sub-sampling (main.py:51,58,60,89)."""
    lo, hi, lerp = O.resize_weights_tf1(60, 61)
    np.testing.assert_array_equal(lo, np.arange(60))
    np.testing.assert_array_equal(hi, np.arange(60) + 1)
    np.testing.assert_allclose(lerp, np.arange(60) / 60.0, atol=4e-6)
    x = np.random.RandomState(2).random_sample((1, 30, 45, 3))
    up = O.resize_bilinear_tf1(x, 60, 90)
    np.testing.assert_array_equal(up[:, 0::2, 0::2], x)
    np.testing.assert_allclose(up[:, 1:59:2, 0::2], 0.5 * (x[:, :-1] + x[:, 1:]), rtol=1e-15)
    np.testing.assert_array_equal(up[:, 59, 0::2], x[:, 29])
    np.testing.assert_array_equal(up[:, 0::2, 89], x[:, :, 44])
    big = np.random.RandomState(3).random_sample((1, 48, 72, 3))
    np.testing.assert_array_equal(O.resize_bilinear_tf1(big, 24, 36), big[:, ::2, ::2])
    np.testing.assert_array_equal(O.resize_bilinear_tf1(big, 12, 18), big[:, ::4, ::4])
    assert O.resize_bilinear_tf1(big, 48, 72) is big


def test_kat2b_resize_23_to_90_columns():
    lo, hi, lerp = O.resize_weights_tf1(90, 23)
    scale = np.float32(23) / np.float32(90)
    src = np.arange(90, dtype=np.float32) * scale
    np.testing.assert_array_equal(lo, np.floor(src).astype(int))
    assert hi.max() == 22 and lo[-1] == 22 and hi[-1] == 22
    assert lerp.dtype == np.float32


def test_kat3_same_padding_stride2_is_1_before_2_after():
    """All-ones 5x5 s2 SAME conv over an all-ones image: top-left sees 4x4=16 taps,
    bottom-right 3x3=9 (pad 1 before / 2 after), interior 25."""
    x = np.ones((1, 480, 720, 1))
    w = np.ones((5, 5, 1, 1))
    y = O.conv2d_same(x, w, 2)[0, :, :, 0]
    assert y.shape == (240, 360)
    assert y[0, 0] == 16 and y[-1, -1] == 9 and y[0, -1] == 12 and y[-1, 0] == 12 and y[100, 100] == 25
    assert O.same_padding(480, 5, 2) == (240, 1, 2)
    assert O.same_padding(90, 9, 1) == (90, 4, 4)
    assert O.same_padding(45, 5, 1) == (45, 2, 2)


def test_kat4_pool_45_to_23_last_column_passes_through():
    x = np.random.RandomState(4).standard_normal((1, 30, 45, 2))
    y = O.max_pool_same(x)
    assert y.shape == (1, 15, 23, 2)
    np.testing.assert_array_equal(y[:, :, 22], np.maximum(x[:, 0::2, 44], x[:, 1::2, 44]))
    np.testing.assert_array_equal(y[:, :, 0], x[:, :, 0:2].reshape(1, 15, 2, 2, 2).max(axis=(2, 3)))
    xt = torch.as_tensor(x).permute(0, 3, 1, 2)
    np.testing.assert_array_equal(T.max_pool_same(xt).permute(0, 2, 3, 1).numpy(), y)


def test_kat5_softmax_sums_to_one_per_map():
    hm = np.random.RandomState(5).standard_normal((2, 60, 90, 9)) * 30
    s = O.spatial_softmax(hm)
    np.testing.assert_allclose(s.sum(axis=(1, 2)), 1.0, rtol=1e-12)
    np.testing.assert_allclose(s, T.spatial_softmax(hm), rtol=1e-10, atol=1e-300)


def test_kat6_kat7_softplus_constants():
    assert abs(O.softplus5(np.float64(0.0)) - np.log(2) / 5) < 1e-15
    assert abs(O.softplus5(np.float64(1e-5)) - (np.log(2) / 5 + 5e-6)) < 1e-10
    x = np.array([-20.0, -13.95, -13.9, -1.0, 0.0, 1.0, 13.9, 13.95, 20.0])
    np.testing.assert_allclose(O.tf_softplus(x), np.logaddexp(0, x), rtol=1e-6)  # exp(x) shortcut below -13.94
    assert O.tf_softplus(np.float64(20.0)) == 20.0


def test_kat8_argmax_tie_takes_lowest_flat_index():
    hm = np.zeros((1, 60, 90, 2))
    hm[0, 10, 20, 0] = hm[0, 10, 21, 0] = hm[0, 40, 3, 0] = 1.0
    hm[0, 59, 89, 1] = 2.0
    c = O.argmax_coords(hm)
    assert c.dtype == np.int32 and c.shape == (1, 2, 2)
    assert tuple(c[0, :, 0]) == (10, 20) and tuple(c[0, :, 1]) == (59, 89)
    assert tuple(O.argmax_coords(np.zeros((1, 60, 90, 1)))[0, :, 0]) == (0, 0)


def test_bn_is_after_relu_and_last_layer_is_linear():
    """main.py:160-165: relu(conv+b) then BN; with negative gamma the order matters."""
    p = {'c/weights': -np.ones((5, 5, 1, 1), np.float32), 'c/biases': np.array([1.0], np.float32),
         'c/BatchNorm/gamma': np.array([-2.0], np.float32), 'c/BatchNorm/beta': np.array([0.5], np.float32),
         'c/BatchNorm/moving_mean': np.array([0.25], np.float32),
         'c/BatchNorm/moving_variance': np.array([3.0], np.float32)}
    x = np.ones((1, 6, 6, 1))
    y = O.conv_layer(x, p, 5, 1, 'c')
    pre = 1.0 - O.conv2d_same(x, np.ones((5, 5, 1, 1)), 1)
    ref = -2.0 * (np.maximum(pre, 0) - 0.25) / np.sqrt(3.0 + 1e-3) + 0.5
    np.testing.assert_allclose(y, ref, rtol=1e-6)
    np.testing.assert_allclose(O.conv_layer(x, p, 5, 1, 'c', last_layer=True), pre, rtol=1e-12)


def test_pair_order_matches_reference_tables():
    assert O.JOINT_DEPENDENCE['lsho'] == ['lelb', 'lwri', 'rsho', 'relb', 'rwri', 'lhip', 'rhip', 'nose', 'torso']
    assert O.JOINT_DEPENDENCE['nose'][-1] == 'torso' and 'nose' not in O.JOINT_DEPENDENCE['nose']
    keys = synth.pair_keys()
    assert len(keys) == 81 and keys[0] == 'lsho_lelb' and keys[-1] == 'nose_torso'


@pytest.mark.parametrize('bn,kind', [('identity', 'init'), ('trained', 'trained')])
def test_two_formulations_agree_end_to_end(bn, kind):
    """numpy-slicing oracle == torch/scipy oracle on the debug-size network at a reduced
    image size (the graph is fully convolutional; 96x144 -> 12x18 maps would break the
    fixed 60x90/120x180 spatial model, so the SM runs on its own 60x90 input)."""
    p = synth.make_pd_params(debug=True, bn=bn, conv6_gain=8.0)
    x = synth.make_images(1, height=120, width=184)        # quarter branch: 15x23 -> exercises pad (0,1)
    a = O.model(x, p)
    b = T.model(x, p)
    assert a.shape == (1, 15, 23, 9)
    np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-9)
    p.update(synth.make_sm_params(synth.synthetic_priors(), kind=kind))
    rs = np.random.RandomState(9)
    hm10 = np.concatenate([O.spatial_softmax(rs.standard_normal((1, 60, 90, 9)) * 3), synth.make_torso(1)], axis=3)
    sa = O.spatial_model(hm10, p)
    sb = T.spatial_model(hm10, p)
    np.testing.assert_allclose(sa, sb, rtol=1e-9, atol=1e-9)
    np.testing.assert_array_equal(O.argmax_coords(O.spatial_softmax(sa)), T.argmax_coords(T.spatial_softmax(sb)))


def test_bf16_round_is_round_to_nearest_even():
    """oracle.bf16_round (the emulate='bf16' mode that pins the bf16 kernels) against torch's float32 -> bfloat16 cast,
    ties included."""
    ties = np.array([1.00390625, 1.01171875, -1.00390625, 3.0e38, 1e-40], np.float64)      # exact halves round to even
    rs = np.random.RandomState(5)
    y = np.concatenate([ties, rs.standard_normal(20000) * 10.0 ** rs.uniform(-30, 30, 20000)])
    want = torch.tensor(y, dtype=torch.float32).to(torch.bfloat16).to(torch.float64).numpy()
    np.testing.assert_array_equal(O.bf16_round(y), want)
    assert O.bf16_round(np.float64(1.00390625)) == 1.0 and O.bf16_round(np.float64(1.01171875)) == 1.015625


def test_emulated_bf16_layer_differs_from_fp32_by_bf16_not_more():
    rs = np.random.RandomState(6)
    p = {'c/weights': (rs.standard_normal((5, 5, 32, 32)) * 0.05).astype(np.float32), 'c/biases': np.zeros(32, np.float32)}
    x = rs.standard_normal((1, 9, 11, 32))
    a = O.conv_layer(x, p, 5, 1, 'c', last_layer=True)
    b = O.conv_layer(x, p, 5, 1, 'c', last_layer=True, emulate='bf16')
    rel = np.abs(a - b).max() / np.abs(a).max()
    assert 1e-4 < rel < 2e-2


@pytest.mark.parametrize('k', [5, 9])
def test_sampled_conv_grads_match_autograd(k):
    """tests/golden_util.sampled_conv_grads (the float64 sums the GPU test of the gradient kernels compares with) against autograd through the
    restated tf.nn.conv2d SAME (oracle/jcm_oracle_torch.conv2d_same): every sampled weight-gradient and data-gradient entry, borders included."""
    from golden_util import sampled_conv_grads
    rs = np.random.RandomState(k)
    B, H, W, cin, cout, lmbd = 2, 7, 10, 3, 4, 0.01
    x = rs.standard_normal((B, H, W, cin)).astype(np.float32)
    dz = rs.standard_normal((B, H, W, cout)).astype(np.float32)
    w = rs.standard_normal((k, k, cin, cout)).astype(np.float32)
    xt = torch.tensor(x, dtype=torch.float64).permute(0, 3, 1, 2).requires_grad_(True)
    wt = torch.tensor(w, dtype=torch.float64).requires_grad_(True)
    z = T.conv2d_same(xt, wt, 1)
    loss = (z * torch.tensor(dz, dtype=torch.float64).permute(0, 3, 1, 2)).sum() + lmbd * (wt ** 2).sum() / 2
    gx, gw = torch.autograd.grad(loss, [xt, wt])
    gx, gw = gx.permute(0, 2, 3, 1).numpy(), gw.numpy().reshape(-1)
    (wi, wv), (xi, xv) = sampled_conv_grads(x, dz, w, lmbd, np.random.RandomState(1), n=60)
    np.testing.assert_allclose(wv, gw[wi], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(xv, [gx[i] for i in xi], rtol=1e-12, atol=1e-12)
