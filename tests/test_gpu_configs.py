"""BASELINE.json configs[2] / configs[3] at their real sizes on the GPU, through size-independent
properties (the float64 oracle needs ~1 s per image, so 256 images are not compared value by value):
every heat map is a distribution, the coordinates are the first-occurrence arg-max of the returned maps,
an image's result does not depend on the batch it travels in (BatchNorm is in inference mode,
main.py:406), and a batch walked in micro-batches equals the same batch in one piece."""
import numpy as np
import pytest
import torch

from golden_util import assert_bf16_coords, batch_golden, config_batch, flic_priors, full_inputs, load, seeds
from joint_cnn_mrf_amd import synth
from oracle import jcm_oracle as O

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32), device='cuda:0')


def _full_params():
    x2, torso2, p = full_inputs()
    p.update(synth.make_sm_params(flic_priors(), kind='trained', seed=seeds()['sm']))
    return x2, torso2, p


@pytest.mark.parametrize('fft', [True, False], ids=['fft', 'mfma'])
def test_config3_batch256_bf16_properties(fft):
    """configs[2] (and a rank's share of configs[3]: 2048 / 8): batch 256, bf16, full-width network,
    FLIC priors.  This is the size at which the arena holds 5 GB, the spatial model runs 16 FFT slices and
    the 9x9 grids have thousands of tiles -- paths a 2-image test never takes."""
    from joint_cnn_mrf_amd.engine import Engine
    x2, torso2, p = _full_params()
    B = 256
    x, torso = config_batch(B)
    eng = Engine(device=0, precision='bf16', conv9_fft=fft).load_params(p)
    # the kernels under test: the frequency-domain route of the wide 9x9 layers (default), or the flattened-strip bf16 MFMA kernel (a
    # silent fallback to the patch kernel would pass too)
    want = 'conv_fft(cgemm_split_kernel)' if fft else 'conv_strip_bf16_kernel'
    assert eng.conv_kernel_name('conv5', B, 60, 90) == eng.conv_kernel_name('conv4_fullres', B, 60, 90) == want
    r = eng.forward(dev(x), dev(torso), use_sm=True)
    pd, sm = r['pd_prob'].cpu().numpy(), r['sm_prob'].cpu().numpy()
    pd_c, sm_c = r['pd_coords'].cpu().numpy(), r['sm_coords'].cpu().numpy()
    assert pd.shape == sm.shape == (B, 60, 90, 9) and pd_c.shape == sm_c.shape == (B, 2, 9)
    np.testing.assert_allclose(pd.sum(axis=(1, 2)), 1.0, rtol=2e-5)
    np.testing.assert_allclose(sm.sum(axis=(1, 2)), 1.0, rtol=2e-5)
    assert np.isfinite(pd).all() and np.isfinite(sm).all() and pd.min() >= 0 and sm.min() >= 0
    np.testing.assert_array_equal(pd_c, O.argmax_coords(pd))
    np.testing.assert_array_equal(sm_c, O.argmax_coords(sm))
    # the first two images are the golden pair: the same engine on a batch of 2 must give the same bits
    # (every output element is one fixed-order MFMA accumulation, whatever tile or workgroup computes it)
    two = eng.forward(dev(x[:2]), dev(torso[:2]), use_sm=True)
    assert np.array_equal(two['pd_prob'].cpu().numpy(), pd[:2])
    np.testing.assert_allclose(two['sm_prob'].cpu().numpy(), sm[:2], atol=1e-6, rtol=0)
    np.testing.assert_array_equal(two['pd_coords'].cpu().numpy(), pd_c[:2])
    np.testing.assert_array_equal(two['sm_coords'].cpu().numpy(), sm_c[:2])
    # an image deep in the batch against a single-image forward
    one = eng.forward(dev(x[201:202]), dev(torso[201:202]), use_sm=True)
    assert np.array_equal(one['pd_prob'].cpu().numpy(), pd[201:202])
    np.testing.assert_allclose(one['sm_prob'].cpu().numpy(), sm[201:202], atol=1e-6, rtol=0)
    np.testing.assert_array_equal(one['sm_coords'].cpu().numpy(), sm_c[201:202])
    # and the golden pair stays within bf16 reach of the float64 oracle (argmax within one cell for most joints)
    assert_bf16_coords(pd_c[:2], load('full_pd_logits'), load('full_pd_coords'), 'pd')
    # ... and so does the LAST image of the batch (tests/golden/batch256.npz: float64 oracle of image 255)
    g = batch_golden(B)
    assert_bf16_coords(pd_c[g['idx']], g['pd_logits'], g['pd_coords'], 'pd')
    eng.close()


def assert_values_vs_golden(pd, sm, pd_c, sm_c, g):
    """The north star's fp32 bar on the stored images of a configuration batch: heat maps within 1e-4 of the float64 oracle's, arg-max
    coordinates identical.  `identical` is asserted for every joint whose golden top-2 margin exceeds the logit bar the other golden tests
    use (2e-4 of the logit scale); below that (a dim image has near-ties) the engine's cell must be one whose golden logit is within that
    bar of the golden maximum -- i.e. an arg-max of a map that IS within tolerance."""
    idx = g['idx']
    for name, got, got_c, ref_l, ref_c, margin in (('pd', pd, pd_c, g['pd_logits'], g['pd_coords'], g['pd_margin']),
                                                    ('sm', sm, sm_c, g['sm_logits'], g['sm_coords'], g['sm_margin'])):
        ref_l = ref_l.astype(np.float64)
        np.testing.assert_allclose(got[idx], O.spatial_softmax(ref_l), atol=1e-4, rtol=0, err_msg=name)
        bar = 2e-4 * max(1.0, float(np.abs(ref_l).max()))
        same = (got_c[idx] == ref_c).all(axis=1)                       # [n, K]
        clear = margin > bar
        assert clear.sum() >= 0.9 * clear.size, (name, int(clear.sum()))
        assert (same | ~clear).all(), (name, np.argwhere(~same & clear).tolist())
        for n, k in np.argwhere(~same):
            r, c = got_c[idx][n, :, k]
            assert ref_l[n, r, c, k] >= ref_l[n, :, :, k].max() - bar, (name, n, k)


def test_config1_batch64_fp32_properties():
    """configs[1], the bench headline: batch 64, fp32, full-width network, FLIC priors, default route (stride-1 layers in the frequency
    domain, channel GEMM on two scaled fp16 parts).  At this size a layer's 64 images share the launches and the GEMM tile: the first two
    images are the golden pair and must come out bit for bit as on a batch of 2 (one power-of-two scale per image, every output element one
    fixed-order accumulation), their arg-max coordinates must be the float64 goldens', and the probabilities must be probabilities."""
    from joint_cnn_mrf_amd.engine import Engine
    x2, torso2, p = _full_params()
    B = 64
    x, torso = config_batch(B)                     # image 7 dim, image 8 bright: their scales differ from their neighbours'
    eng = Engine(device=0).load_params(p)
    assert eng.conv_kernel_name('conv5', B, 60, 90) == eng.conv_kernel_name('conv2_fullres', B, 120, 180) == 'conv_fft(cgemm_split_kernel)'
    r = eng.forward(dev(x), dev(torso), use_sm=True)
    pd, sm = r['pd_prob'].cpu().numpy(), r['sm_prob'].cpu().numpy()
    pd_c, sm_c = r['pd_coords'].cpu().numpy(), r['sm_coords'].cpu().numpy()
    assert pd.shape == sm.shape == (B, 60, 90, 9) and pd_c.shape == sm_c.shape == (B, 2, 9)
    np.testing.assert_allclose(pd.sum(axis=(1, 2)), 1.0, rtol=2e-5)
    np.testing.assert_allclose(sm.sum(axis=(1, 2)), 1.0, rtol=2e-5)
    assert np.isfinite(pd).all() and np.isfinite(sm).all() and pd.min() >= 0 and sm.min() >= 0
    np.testing.assert_array_equal(pd_c, O.argmax_coords(pd))
    np.testing.assert_array_equal(sm_c, O.argmax_coords(sm))
    np.testing.assert_array_equal(pd_c[:2], load('full_pd_coords'))
    np.testing.assert_array_equal(sm_c[:2], load('full_sm_coords_trained'))
    np.testing.assert_allclose(pd[:2], O.spatial_softmax(load('full_pd_logits').astype(np.float64)), atol=1e-4, rtol=0)
    np.testing.assert_allclose(sm[:2], O.spatial_softmax(load('full_sm_logits_trained').astype(np.float64)), atol=1e-4, rtol=0)
    # VALUE comparison at eight more positions of the batch (dim, bright, middle, both sides of row 32, the end): tests/golden/batch64.npz
    assert_values_vs_golden(pd, sm, pd_c, sm_c, batch_golden(B))
    two = eng.forward(dev(x[:2]), dev(torso[:2]), use_sm=True)
    assert np.array_equal(two['pd_prob'].cpu().numpy(), pd[:2])
    np.testing.assert_allclose(two['sm_prob'].cpu().numpy(), sm[:2], atol=1e-6, rtol=0)
    for i in (7, 8, 41):                           # the dim image, the bright one, one deep in the batch: alone = in the batch
        one = eng.forward(dev(x[i:i + 1]), dev(torso[i:i + 1]), use_sm=True)
        assert np.array_equal(one['pd_prob'].cpu().numpy(), pd[i:i + 1]), i
        np.testing.assert_array_equal(one['sm_coords'].cpu().numpy(), sm_c[i:i + 1])
    eng.close()


def test_config3_global_batch_2048_bf16_micro_batched():
    """configs[3] at its single-GPU size: 2048 images in ONE jcm_forward, walked in 8 micro-batches of 256 (what `bench.py --global-batch 2048`
    times on one rank, main.py:511-517).  2048 x 11 frequency-domain layers would lap a fixed ring of fp16 scale words on an fp32 handle; on
    this bf16 handle the call exercises the arena reuse across micro-batches and the output offsets.  Properties: every map a distribution,
    coordinates = first-occurrence arg-max of the returned maps, the golden pair (images 0, 1) bit-identical to a batch of 2, image 1500 (row
    220 of micro-batch 5) alone = in the batch, the last image (the end of the last micro-batch) alone = in the batch."""
    from joint_cnn_mrf_amd.engine import Engine
    x2, torso2, p = _full_params()
    B = 2048
    g = torch.Generator(device='cuda:0')
    g.manual_seed(2048)
    x = torch.rand((B, 480, 720, 3), device='cuda:0', generator=g)        # 8.5 GB: generated on the device (U[0,1) like data.py:129-130)
    x[:2] = dev(x2)
    torso = dev(np.concatenate([torso2, synth.make_torso(B - 2, seed=2049)], axis=0))
    eng = Engine(device=0, precision='bf16').load_params(p)
    assert eng.conv_kernel_name('conv5', 256, 60, 90) == 'conv_fft(cgemm_split_kernel)'
    r = eng.forward(x, torso, use_sm=True)
    pd, sm, pd_c, sm_c = r['pd_prob'], r['sm_prob'], r['pd_coords'], r['sm_coords']
    assert tuple(pd.shape) == tuple(sm.shape) == (B, 60, 90, 9) and tuple(sm_c.shape) == (B, 2, 9)
    for t in (pd, sm):
        s = t.sum(dim=(1, 2))
        assert bool(torch.isfinite(t).all()) and float(t.min()) >= 0 and float((s - 1).abs().max()) <= 2e-5
    for t, cc in ((pd, pd_c), (sm, sm_c)):                       # first-occurrence arg-max of the returned maps (evaluation.py:15-24), sampled
        for i in (0, 255, 256, 1023, 1500, 2047):
            np.testing.assert_array_equal(cc[i:i + 1].cpu().numpy(), O.argmax_coords(t[i:i + 1].cpu().numpy()))
    for sl in (slice(0, 2), slice(1500, 1501), slice(2047, 2048)):
        one = eng.forward(x[sl].contiguous(), torso[sl].contiguous(), use_sm=True)
        assert torch.equal(one['pd_prob'], pd[sl]), sl
        assert float((one['sm_prob'] - sm[sl]).abs().max()) <= 1e-6
        assert torch.equal(one['pd_coords'], pd_c[sl]) and torch.equal(one['sm_coords'], sm_c[sl]), sl
    assert_bf16_coords(pd_c[:2].cpu().numpy(), load('full_pd_logits'), load('full_pd_coords'), 'pd')
    eng.close()


def test_fp32_forward_beyond_one_block_of_scale_words():
    """An fp32 handle takes one fp16 scale word per image and frequency-domain layer (csrc/ctx.h: kFftWords = 2^18 per block).  A single
    jcm_pd_forward of more images than one block serves must get further blocks instead of failing mid-call (round 3 returned JCM_ERR_STATE
    above ~5900 images) -- checked at --debug width on 32x48 images so that it runs in seconds: the five frequency-domain layers of that network (conv4_*, conv5 and
    its hand-over to conv6) take 5 words per image, 56000 images 280000 > 2^18 -- every
    image's logits equal to the same image's in a batch of 3, and the next call starts over at the first block."""
    from joint_cnn_mrf_amd.engine import Engine
    p = synth.make_pd_params(debug=True, bn='trained', conv6_gain=8.0)
    eng = Engine(device=0).load_params(p)
    B = 56000
    g = torch.Generator(device='cuda:0')
    g.manual_seed(7)
    x = torch.rand((B, 32, 48, 3), device='cuda:0', generator=g)
    assert eng.conv_kernel_name('conv5', B, 4, 6).startswith('conv_fft')
    big = eng.model(x)
    for i in (0, 12345, 52500, B - 1):
        small = eng.model(x[i:i + 1].contiguous())
        assert torch.equal(small[0], big[i]), i
    again = eng.model(x)
    assert torch.equal(again, big)
    eng.close()


@pytest.mark.parametrize('precision,debug,B,mb', [('fp32', True, 5, 2), ('bf16', False, 5, 2), ('fp32', False, 3, 2)])
def test_micro_batched_forward_equals_one_piece(precision, debug, B, mb):
    """jcm_forward walks a large batch in micro-batches (option "micro_batch": a rank's 2048 / N share of
    configs[3]); ragged last slice included, every output lands at its image offset, results identical."""
    from joint_cnn_mrf_amd.engine import Engine
    p = synth.make_pd_params(debug=debug, bn='trained', conv6_gain=8.0)
    p.update(synth.make_sm_params(synth.synthetic_priors(), kind='trained'))
    x, torso = dev(synth.make_images(B, seed=5)), dev(synth.make_torso(B, seed=6))
    eng = Engine(device=0, precision=precision, micro_batch=B).load_params(p)
    whole = eng.forward(x, torso, use_sm=True)
    whole = {k: v.clone() for k, v in whole.items()}
    eng.set_micro_batch(mb)
    parts = eng.forward(x, torso, use_sm=True)
    for k in ('pd_prob', 'pd_coords', 'sm_coords'):
        assert torch.equal(parts[k], whole[k]), k
    assert float((parts['sm_prob'] - whole['sm_prob']).abs().max()) <= 1e-6
    # coordinates only (the bench's call): no probability outputs requested
    c = eng.forward(x, torso, use_sm=True, want_prob=False)
    assert torch.equal(c['sm_coords'], whole['sm_coords']) and torch.equal(c['pd_coords'], whole['pd_coords'])
    eng.close()


def test_fused_softmax_argmax_matches_oracle_and_ties():
    """The tail of the tower (spatial_softmax + arg-max in one kernel) against the float64 oracle, with
    exact ties: the FIRST flat index wins (np.argmax / tf.argmax, KAT8)."""
    from joint_cnn_mrf_amd.engine import Engine
    eng = Engine(device=0)
    eng.finalize()
    rs = np.random.RandomState(4)
    z = (3.0 * rs.standard_normal((5, 60, 90, 9))).astype(np.float32)
    z[0, :, :, 0] = 0.25                       # a constant map: every pixel ties -> (0, 0)
    z[1, 17, 40, 3] = z[1, 52, 7, 3] = 40.0    # two equal maxima: the earlier flat index (17, 40) wins
    z[2, 59, 89, 8] = 50.0                     # the very last pixel
    z[3, 0, 0, 5] = 50.0                       # the very first
    prob, coords = eng.softmax_argmax(dev(z))
    ref = O.spatial_softmax(z.astype(np.float64))
    np.testing.assert_allclose(prob.cpu().numpy(), ref, rtol=2e-5, atol=1e-12)
    c = coords.cpu().numpy()
    np.testing.assert_array_equal(c, O.argmax_coords(prob.cpu().numpy()))
    assert tuple(c[0, :, 0]) == (0, 0) and tuple(c[1, :, 3]) == (17, 40) and tuple(c[2, :, 8]) == (59, 89) and tuple(c[3, :, 5]) == (0, 0)
    _none, coords_only = eng.softmax_argmax(dev(z), want_prob=False)
    assert torch.equal(coords_only, coords)
    # other joint counts / map sizes take the general two-kernel route behind the same entry point
    z2 = rs.standard_normal((2, 15, 23, 4)).astype(np.float32)
    p2, c2 = eng.softmax_argmax(dev(z2))
    np.testing.assert_allclose(p2.cpu().numpy(), O.spatial_softmax(z2.astype(np.float64)), rtol=2e-5, atol=1e-12)
    np.testing.assert_array_equal(c2.cpu().numpy(), O.argmax_coords(p2.cpu().numpy()))
    eng.close()


def test_coords_only_argmax_equals_argmax_of_probabilities_under_near_ties():
    """The coordinates-only route of the fused tail skips the division for every pixel whose exp is not within a few ulp of
    1.  Maps whose two largest logits differ by 0, 1, 2 and 3 float32 ulps (the later pixel holding the larger one) are
    where a rounded quotient can tie: both routes must return the same pixel as np.argmax of the returned probabilities."""
    from joint_cnn_mrf_amd.engine import Engine
    eng = Engine(device=0)
    eng.finalize()
    rs = np.random.RandomState(8)
    z = (2.0 * rs.standard_normal((40, 60, 90, 9))).astype(np.float32)
    top = np.float32(9.0)
    for b in range(40):
        for k in range(9):
            lo, hi = sorted(rs.choice(5400, 2, replace=False))
            later = top
            for _ in range((b + k) % 4):
                later = np.nextafter(later, np.float32(np.inf))
            z[b].reshape(5400, 9)[lo, k] = top          # earlier pixel: the smaller (or equal) logit
            z[b].reshape(5400, 9)[hi, k] = later
    prob, coords = eng.softmax_argmax(dev(z))
    _none, coords_only = eng.softmax_argmax(dev(z), want_prob=False)
    eng.close()
    np.testing.assert_array_equal(coords.cpu().numpy(), O.argmax_coords(prob.cpu().numpy()))
    assert torch.equal(coords_only, coords)
