"""Parity of the HIP path (through the C ABI) against the float64 oracle on the same seeded
inputs, at sizes the oracle finishes in seconds (debug-size filters, main.py:40-41).

Tolerances (north star): heat maps |d| <= 1e-4 in fp32, argmax coordinates bit-exact.
1e-4 on softmax outputs (values <= 1, typically ~1e-3) is loose, so logits are additionally
held to 2e-4 * max(1, max|logit|)."""
import numpy as np
import pytest
import torch

import joint_cnn_mrf_amd  # noqa: F401
from joint_cnn_mrf_amd import synth
from oracle import jcm_oracle as O

pytestmark = pytest.mark.gpu

HM_TOL = 1e-4


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32), device='cuda:0')


def logit_tol(ref):
    return 2e-4 * max(1.0, float(np.abs(ref).max()))


@pytest.fixture(scope='module')
def debug_setup():
    from joint_cnn_mrf_amd.engine import Engine
    p = synth.make_pd_params(debug=True, bn='trained', conv6_gain=8.0)
    p.update(synth.make_sm_params(synth.synthetic_priors(), kind='trained'))
    eng = Engine(device=0).load_params(p)
    yield eng, p
    eng.close()


def test_conv1_stride2_asymmetric_padding(debug_setup):
    eng, p = debug_setup
    x = synth.make_images(2, seed=3)
    for res, sub in (('fullres', 1), ('halfres', 2), ('quarterres', 4)):
        xin = x[:, ::sub, ::sub]
        ref = O.conv_layer(xin, p, 5, 2, 'conv1_' + res)
        got = eng.conv_layer(dev(xin), 'conv1_' + res, 2, n_out=16).cpu().numpy()
        assert got.shape == ref.shape
        np.testing.assert_allclose(got, ref, atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize('scope,size,hw', [
    ('conv2_fullres', 5, (120, 180)), ('conv3_fullres', 5, (60, 90)), ('conv4_fullres', 9, (60, 90)),
    ('conv3_halfres', 5, (30, 45)), ('conv4_halfres', 9, (30, 45)), ('conv4_quarterres', 9, (15, 23)),
    ('conv5', 9, (60, 90)),
])
def test_conv_layers_igemm(debug_setup, scope, size, hw):
    eng, p = debug_setup
    cin, cout = p[scope + '/weights'].shape[2:]
    x = np.random.RandomState(hash(scope) % 1000).standard_normal((2, hw[0], hw[1], cin)).astype(np.float32)
    ref = O.conv_layer(x, p, size, 1, scope)
    got = eng.conv_layer(dev(x), scope, 1, n_out=cout).cpu().numpy()
    np.testing.assert_allclose(got, ref, atol=1e-4, rtol=1e-4)


def test_conv6_last_layer_is_linear(debug_setup):
    eng, p = debug_setup
    x = np.random.RandomState(6).standard_normal((2, 60, 90, 128)).astype(np.float32)
    ref = O.conv_layer(x, p, 9, 1, 'conv6', last_layer=True)
    got = eng.conv_layer(dev(x), 'conv6', 1, last_layer=True, n_out=9).cpu().numpy()
    np.testing.assert_allclose(got, ref, atol=logit_tol(ref), rtol=0)
    with pytest.raises(RuntimeError, match='last_layer'):
        eng.conv_layer(dev(x), 'conv6', 1, last_layer=False, n_out=9)


def test_max_pool_and_resize(debug_setup):
    eng, _ = debug_setup
    x = np.random.RandomState(7).standard_normal((2, 30, 45, 32)).astype(np.float32)
    np.testing.assert_array_equal(eng.max_pool(dev(x)).cpu().numpy(), O.max_pool_same(x))
    x2 = np.random.RandomState(8).standard_normal((2, 240, 360, 16)).astype(np.float32)
    np.testing.assert_array_equal(eng.max_pool(dev(x2)).cpu().numpy(), O.max_pool_same(x2))
    for (h, w, c, oh, ow) in [(30, 45, 32, 60, 90), (15, 23, 32, 60, 90), (61, 91, 1, 60, 90), (48, 72, 3, 24, 36), (60, 90, 8, 60, 90)]:
        xr = np.random.RandomState(h).standard_normal((2, h, w, c)).astype(np.float32)
        got = eng.resize_bilinear(dev(xr), oh, ow).cpu().numpy()
        np.testing.assert_allclose(got, O.resize_bilinear_tf1(xr, oh, ow), atol=1e-6, rtol=0)   # fp32 both sides, same expression order (built with -ffp-contract=off)
        np.testing.assert_allclose(got, O.resize_bilinear_tf1(xr.astype(np.float64), oh, ow), atol=1e-5, rtol=0)


def test_model_fused_equals_layerwise_and_oracle(debug_setup):
    eng, p = debug_setup
    from joint_cnn_mrf_amd import main as M
    M._engine, M.hps.debug = eng, True
    x = synth.make_images(2, seed=21)
    ref = O.model(x, p)
    fused = M.model(dev(x), 9).cpu().numpy()
    layer = M.model_layerwise(dev(x), 9).cpu().numpy()
    M._engine = None
    assert fused.shape == (2, 60, 90, 9)
    np.testing.assert_allclose(fused, ref, atol=logit_tol(ref), rtol=0)
    np.testing.assert_allclose(layer, fused, atol=logit_tol(ref), rtol=0)   # same kernels, different fp32 association in the merge


def test_spatial_softmax_and_argmax(debug_setup):
    eng, _ = debug_setup
    hm = (np.random.RandomState(10).standard_normal((3, 60, 90, 9)) * 6).astype(np.float32)
    got = eng.spatial_softmax(dev(hm)).cpu().numpy()
    ref = O.spatial_softmax(hm.astype(np.float64))
    np.testing.assert_allclose(got, ref, atol=1e-7, rtol=2e-5)
    np.testing.assert_allclose(got.sum(axis=(1, 2)), 1.0, rtol=1e-5)
    np.testing.assert_array_equal(eng.argmax_coords(dev(hm)).cpu().numpy(), O.argmax_coords(hm))
    tie = np.zeros((1, 60, 90, 2), np.float32)
    tie[0, 10, 20, 0] = tie[0, 10, 21, 0] = tie[0, 40, 3, 0] = 1.0
    tie[0, 59, 89, 1] = 2.0
    c = eng.argmax_coords(dev(tie)).cpu().numpy()
    assert c.dtype == np.int32 and tuple(c[0, :, 0]) == (10, 20) and tuple(c[0, :, 1]) == (59, 89)


@pytest.mark.parametrize('algo', ['fft', 'direct'])
def test_conv_mrf_delta_and_random(debug_setup, algo):
    eng, _ = debug_setup
    eng.set_sm_algo(algo)
    rs = np.random.RandomState(12)
    A = rs.random_sample((1, 120, 180, 1)).astype(np.float32)
    Bm = np.zeros((3, 60, 90, 1), np.float32)
    Bm[0, 0, 0, 0] = Bm[1, 59, 89, 0] = Bm[2, 17, 42, 0] = 1.0          # KAT1
    got = eng.conv_mrf(dev(A), dev(Bm)).cpu().numpy()
    np.testing.assert_allclose(got, O.conv_mrf(A.astype(np.float64), Bm.astype(np.float64)), atol=2e-6, rtol=0)
    Br = rs.random_sample((2, 60, 90, 1)).astype(np.float32)
    ref = O.conv_mrf(A.astype(np.float64), Br.astype(np.float64))
    got = eng.conv_mrf(dev(A), dev(Br)).cpu().numpy()
    eng.set_sm_algo('fft')
    np.testing.assert_allclose(got, ref, rtol=2e-5, atol=0)   # 5400-term fp32 sums / fp32 FFT


@pytest.mark.parametrize('algo', ['fft_fused', 'direct'])
@pytest.mark.parametrize('kind', ['init', 'trained'])
def test_spatial_model(kind, algo):
    from joint_cnn_mrf_amd.engine import Engine
    p = synth.make_sm_params(synth.synthetic_priors(), kind=kind)
    eng = Engine(device=0).load_params(p)
    eng.set_sm_algo(algo)
    rs = np.random.RandomState(14)
    hm10 = np.concatenate([O.spatial_softmax(rs.standard_normal((2, 60, 90, 9)) * 4), synth.make_torso(2)], axis=3).astype(np.float32)
    ref = O.spatial_model(hm10.astype(np.float64), p)
    got = eng.spatial_model(dev(hm10)).cpu().numpy()
    eng.close()
    np.testing.assert_allclose(got, ref, atol=logit_tol(ref), rtol=0)
    np.testing.assert_allclose(O.spatial_softmax(got.astype(np.float64)), O.spatial_softmax(ref), atol=HM_TOL, rtol=0)


def test_full_tower_debug_size(debug_setup):
    """Config-2 shape at debug width: part detector + spatial model + argmax, B=4."""
    eng, p = debug_setup
    x, torso = synth.make_images(4, seed=31), synth.make_torso(4, seed=32)
    ref = O.forward(x, torso, p)
    r = eng.forward(dev(x), dev(torso), use_sm=True)
    for k in ('pd_prob', 'sm_prob'):
        np.testing.assert_allclose(r[k].cpu().numpy(), ref[k], atol=HM_TOL, rtol=0)
        np.testing.assert_allclose(r[k].cpu().numpy(), ref[k], rtol=2e-3, atol=1e-9)
    for k in ('pd_coords', 'sm_coords'):
        np.testing.assert_array_equal(r[k].cpu().numpy(), ref[k])
    r2 = eng.forward(dev(x), None, use_sm=False, want_prob=False)
    np.testing.assert_array_equal(r2['pd_coords'].cpu().numpy(), ref['pd_coords'])
    assert 'sm_coords' not in r2


def test_shim_validates_before_the_call(debug_setup):
    eng, _ = debug_setup
    with pytest.raises(ValueError):
        eng.model(torch.zeros(1, 480, 720, 4, device='cuda:0'))
    with pytest.raises(TypeError):
        eng.model(torch.zeros(1, 480, 720, 3, device='cuda:0', dtype=torch.float16))
    with pytest.raises(ValueError):
        eng.model(torch.zeros(1, 3, 480, 720, device='cuda:0').permute(0, 2, 3, 1))
    with pytest.raises(ValueError):
        eng.forward(torch.zeros(1, 480, 720, 3, device='cuda:0'), None, use_sm=True)


def test_merged_layer_entry_point_matches_the_tower(debug_setup):
    """jcm_conv_layer_merged = main.py:58,67,69-71 on the branch outputs of the oracle's own forward: conv5 of ((x1 + up(x2)) + up(x3)) / 3."""
    eng, p = debug_setup
    x = synth.make_images(2, seed=7)
    taps = {}
    O.model(x, p, taps=taps)
    x1, x2, x3 = (dev(taps['conv4_' + r].astype(np.float32)) for r in ('fullres', 'halfres', 'quarterres'))
    got = eng.conv_layer_merged(x1, x2, x3, 'conv5', taps['conv5'].shape[-1]).cpu().numpy()
    np.testing.assert_allclose(got, taps['conv5'], atol=2e-5 * np.abs(taps['conv5']).max(), rtol=0)
    with pytest.raises(RuntimeError, match='BatchNorm'):
        eng.conv_layer_merged(x1, x2, x3, 'conv6', 9)             # the logits layer has no BatchNorm: not a layer the merge feeds
    with pytest.raises(RuntimeError, match='no conv layer'):
        eng.conv_layer_merged(x1, x2, x3, 'conv7', 9)
    with pytest.raises(ValueError):
        eng.conv_layer_merged(x1, x2[:1], x3, 'conv5', 128)       # batch mismatch is caught in the binding


@pytest.mark.parametrize('gain', [1e-5, 255.0, 3e5])
def test_tower_is_range_free_in_the_image(debug_setup, gain):
    """The default fp32 route feeds conv1 as two FP16 parts of the window times ITS OWN power of two (conv1_mfma.hip) and the stride-1 layers as scaled
    fp16 parts of their spectra: images in [0, 1e-5], [0, 255] or [0, 3e5] go through like images in [0, 1]."""
    eng, p = debug_setup
    x = (synth.make_images(2, seed=13) * np.float32(gain)).astype(np.float32)
    ref = O.model(x, p)
    got = eng.model(dev(x)).cpu().numpy()
    assert np.isfinite(got).all()
    np.testing.assert_allclose(got, ref, atol=logit_tol(ref), rtol=0)


def test_other_resolution_and_batch_shapes(debug_setup):
    """The part detector is fully convolutional (main.py:34 only documents 480x720): a 240x368
    image gives 30x46 maps, the quarter branch runs on 60x92 -> 8x12 with the SAME pool padding
    in play; batch 1 and an odd batch go through the same kernels."""
    eng, p = debug_setup
    x = synth.make_images(3, seed=41, height=240, width=368)
    ref = O.model(x, p)
    got = eng.model(dev(x)).cpu().numpy()
    assert got.shape == ref.shape == (3, 30, 46, 9)
    np.testing.assert_allclose(got, ref, atol=logit_tol(ref), rtol=0)
    one = eng.model(dev(x[:1])).cpu().numpy()
    np.testing.assert_allclose(one, got[:1], atol=1e-6, rtol=0)       # batch-size independent (no cross-image term)
    r = eng.forward(dev(x), None, use_sm=False)
    np.testing.assert_array_equal(r['pd_coords'].cpu().numpy(), O.argmax_coords(O.spatial_softmax(ref)))
    with pytest.raises(RuntimeError, match='60x90'):
        eng.forward(dev(x), dev(synth.make_torso(3)), use_sm=True)      # the spatial model is 60x90-only


def test_missing_parameters_fail_loudly():
    from joint_cnn_mrf_amd.engine import Engine
    p = synth.make_pd_params(debug=True)
    del p['conv5/biases']
    with pytest.raises(RuntimeError, match='conv5/biases'):
        Engine(device=0).load_params(p)
    eng = Engine(device=0).load_params(synth.make_pd_params(debug=True))
    with pytest.raises(RuntimeError, match='spatial-model parameters'):
        eng.spatial_model(torch.zeros(1, 60, 90, 10, device='cuda:0'))
    with pytest.raises(RuntimeError, match='finalize'):
        Engine(device=0).model(torch.zeros(1, 480, 720, 3, device='cuda:0'))
    sm = synth.make_sm_params(synth.synthetic_priors())
    del sm['energy_nose_torso']
    with pytest.raises(RuntimeError, match='energy_nose_torso'):
        Engine(device=0).load_params(sm)


@pytest.mark.parametrize('f32_conv', ['split16'])
def test_split_convs_debug_width_and_odd_shapes(f32_conv):
    """f32_conv='split16' (conv_split.hip, two fp16 parts per operand) on the shapes the other tests use: at --debug width the 128-channel
    layers take the 128-channel tiles, 30x45 / 15x23 maps the whole-row tiles, small grids fall back to the exact
    kernel -- same tolerances as the exact path, including a non-480x720 input and batch 1."""
    from joint_cnn_mrf_amd.engine import Engine
    p = synth.make_pd_params(debug=True, bn='trained', conv6_gain=8.0)
    p.update(synth.make_sm_params(synth.synthetic_priors(), kind='trained'))
    eng = Engine(device=0, f32_conv=f32_conv, split_min_wgs=0).load_params(p)
    x, torso = synth.make_images(4, seed=31), synth.make_torso(4, seed=32)
    ref = O.forward(x, torso, p)
    r = eng.forward(dev(x), dev(torso), use_sm=True)
    for k in ('pd_prob', 'sm_prob'):
        np.testing.assert_allclose(r[k].cpu().numpy(), ref[k], atol=HM_TOL, rtol=0)
    for k in ('pd_coords', 'sm_coords'):
        np.testing.assert_array_equal(r[k].cpu().numpy(), ref[k])
    x2 = synth.make_images(3, seed=41, height=240, width=368)
    ref2 = O.model(x2, p)
    got2 = eng.model(dev(x2)).cpu().numpy()
    np.testing.assert_allclose(got2, ref2, atol=logit_tol(ref2), rtol=0)
    one = eng.model(dev(x2[:1])).cpu().numpy()
    np.testing.assert_allclose(one, got2[:1], atol=logit_tol(ref2), rtol=0)   # a different batch may pick a different kernel
    eng.close()


@pytest.mark.parametrize('B', [30, 37, 64])
def test_spatial_model_balanced_cuts_are_bit_identical(debug_setup, B):
    """sm_inv_finish_kernel cuts the (image, joint, pair) units into equal ranges when the items do not fill whole rounds of the resident work groups
    (B = 30, 37, 64: 270 / 333 / 576 items on 256 CUs); a range that ends inside an item publishes the partial sums of its head and the next range
    continues them IN GRAPH ORDER (sm_fused.hip).  Every logit must therefore equal, bit for bit, the logit of the same image run alone (one work group
    per item, no cut) -- and the oracle."""
    eng, p = debug_setup
    rs = np.random.RandomState(100 + B)
    hm = (rs.random_sample((B, 60, 90, 10)) ** 8 * 0.02).astype(np.float32)
    got = eng.spatial_model(dev(hm)).cpu().numpy()
    for b in (0, 1, B // 2, B - 2, B - 1):
        one = eng.spatial_model(dev(hm[b:b + 1])).cpu().numpy()
        assert np.array_equal(one[0], got[b]), b
    again = eng.spatial_model(dev(hm)).cpu().numpy()      # the flags carry the launch epoch: a second launch must not see the first one's
    assert np.array_equal(again, got)
    ref = O.spatial_model(hm[:2].astype(np.float64), p)
    np.testing.assert_allclose(got[:2], ref, atol=logit_tol(ref), rtol=0)
