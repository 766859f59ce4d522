"""Shape sweep of the stride-1 convolution kernels beyond the model's own maps: seeded random (batch, height, width,
channels, kernel size) cases through jcm_conv_layer, on the exact fp32 MFMA path, the two split paths (forced onto
small grids) and the bf16 path, against the float64 oracle.  Catches tile-edge / padding / channel-tile mistakes the
fixed 60x90-family shapes cannot."""
import numpy as np
import pytest
import torch

import joint_cnn_mrf_amd  # noqa: F401
from oracle import jcm_oracle as O

pytestmark = pytest.mark.gpu


def cases():
    rs = np.random.RandomState(2024)
    out = []
    for i in range(14):
        ks = int(rs.choice([5, 9]))
        cin = int(rs.choice([32, 64, 96]))
        cout = int(rs.choice([32, 64, 128, 256]))
        out.append((i, int(rs.choice([1, 2, 5])), int(rs.randint(6, 70)), int(rs.randint(9, 130)), cin, cout, ks))
    out.append((14, 3, 24, 64, 64, 128, 9))      # 12x32-patch path of the split kernels (W >= 64, H % 12 == 0)
    out.append((15, 2, 36, 96, 32, 256, 5))
    # conv5_strip_bf16_kernel (5x5, 128 output channels, 768-position strips): the model's own maps, a last tile of 1, 2 and 3 fragment
    # rows per wave, three row parts (W > 128), one (W < 64); 15x23 is too narrow for its 64-entry window table -> patch kernel
    out += [(16, 2, 60, 90, 128, 128, 5), (17, 1, 120, 180, 64, 128, 5), (18, 3, 30, 45, 64, 128, 5), (19, 2, 15, 23, 128, 128, 5),
            (20, 5, 17, 29, 32, 128, 5), (21, 1, 9, 128, 96, 128, 5)]
    # conv_fft (fp32, wide 9x9 layers in the frequency domain): the model's three map sizes (72x100, 40x60, 24x32 transforms) and a map
    # that fills its 72x100 transform to the last row and column
    out += [(22, 2, 60, 90, 256, 512, 9), (23, 1, 30, 45, 128, 128, 9), (24, 3, 15, 23, 128, 256, 9), (25, 1, 64, 92, 128, 128, 9),
            (26, 1, 120, 180, 64, 128, 5), (27, 2, 97, 121, 64, 64, 9)]      # 128 x 192 transforms (32-channel column blocks)
    return out


def layer_params(rs, cin, cout, ks):
    return {'c/weights': (rs.standard_normal((ks, ks, cin, cout)) * np.sqrt(2.0 / (ks * ks * cin))).astype(np.float32),
            'c/biases': (0.1 * rs.standard_normal(cout)).astype(np.float32),
            'c/BatchNorm/gamma': rs.uniform(0.5, 1.5, cout).astype(np.float32), 'c/BatchNorm/beta': (0.1 * rs.standard_normal(cout)).astype(np.float32),
            'c/BatchNorm/moving_mean': (0.1 * rs.standard_normal(cout)).astype(np.float32),
            'c/BatchNorm/moving_variance': rs.uniform(0.5, 1.5, cout).astype(np.float32)}


@pytest.mark.parametrize('case', cases(), ids=lambda c: 'B%d_%dx%d_%d-%d_k%d' % c[1:])
def test_conv_layer_random_shape(case):
    from joint_cnn_mrf_amd.engine import Engine
    i, B, H, W, cin, cout, ks = case
    rs = np.random.RandomState(100 + i)
    p = layer_params(rs, cin, cout, ks)
    x = rs.standard_normal((B, H, W, cin)).astype(np.float32)
    ref = O.conv_layer(x.astype(np.float64), p, ks, 1, 'c')
    scale = np.abs(ref).max()
    xd = torch.as_tensor(x, device='cuda:0')
    for mode, kw, tol in (('exact', dict(f32_conv='exact'), 2e-5), ('chain', dict(f32_conv='exact', conv9_fft=False), 2e-5),
                          ('split16', dict(f32_conv='split16', split_min_wgs=0), 2e-5)):
        eng = Engine(device=0, **kw).load_params(p)
        if mode in ('exact', 'chain'):      # 'exact' = the default fp32 engine: frequency domain whenever the shape allows
            assert eng.conv_kernel_name('c', B, H, W).startswith('conv_fft') == (mode == 'exact' and cin % 64 == 0 and H + ks <= 193 and W + ks <= 193)
        got = eng.conv_layer(xd, 'c', 1, n_out=cout).cpu().numpy()
        eng.close()
        err = np.abs(got - ref).max() / scale
        assert err <= tol, '%s: %.2e' % (mode, err)
    # bf16 path, per layer: against the oracle in ITS arithmetic (operands rounded to bf16, wide accumulation, result
    # rounded to bf16).  The kernel accumulates in fp32, so a result that lands on a rounding boundary may come out one
    # bf16 ulp away; everything else must be identical.
    refb = O.conv_layer(x.astype(np.float64), p, ks, 1, 'c', emulate='bf16')
    eng = Engine(device=0, precision='bf16').load_params(p)
    if 16 <= i <= 21:
        assert eng.conv_kernel_name("c", B, H, W) == "conv5_strip_bf16_kernel"
    freq = eng.conv_kernel_name('c', B, H, W).startswith('conv_fft')
    gotb = eng.conv_layer(xd, 'c', 1, n_out=cout).cpu().numpy().astype(np.float64)
    eng.close()
    if not freq:
        check_bf16_layer(gotb, refb)
        return
    # A wide 9x9 layer of a bf16 handle runs in the frequency domain.  Default (round 4): ONE fp16 part per spectrum and 16-bit row-transformed
    # tensors (fft_single, fft_t16: 11 significant bits at every intermediate, the layer's own tensors have 8) -- the looser bar of
    # test_bf16_tower_vs_bf16_oracle; with both off (two bf16 parts per spectrum, fp32 row-transformed tensors) the strict one-ulp bar holds.
    check_bf16_layer(gotb, refb, slack_rel=1e-3, flips=0.12, rms_rel=4e-4)
    eng = Engine(device=0, precision='bf16', fft_single=False).load_params(p)
    gotb = eng.conv_layer(xd, 'c', 1, n_out=cout).cpu().numpy().astype(np.float64)
    eng.close()
    check_bf16_layer(gotb, refb)


def check_bf16_layer(got, ref, slack_rel=1e-5, flips=0.02, rms_rel=1e-3):
    """got, ref: bf16-valued arrays.  At most one bf16 ulp apart (plus the accumulation error of the kernel, which
    is relative to the layer's scale, not to a result that the bias / BatchNorm shift happens to bring near zero), and
    apart in at most 2 % of the entries.  (The looser arguments are for the frequency-domain route with 11-bit spectra, test_gpu_golden.py.)"""
    ulp = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(ref), 1e-30))) - 7)
    diff = np.abs(got - ref)
    slack = slack_rel * np.abs(ref).max()
    assert (diff <= 1.001 * ulp + slack).all(), 'more than one bf16 ulp: worst %.3g ulp' % float(((diff - slack) / ulp).max())
    assert (diff > 0).mean() <= flips, 'rounded differently in %.2f %% of the entries' % (100 * (diff > 0).mean())
    assert np.sqrt(np.mean(diff ** 2)) <= rms_rel * np.abs(ref).max()


# 82 <= W <= 90 with 9 joints: conv_kxfold_bf16_kernel (kernel columns folded into N); its last tile of an image runs 3, 2 or 1
# fragments per wave: 60x90 -> 2, 9x82 -> 1, 23x86 -> 3; 61x88: odd height, rows that straddle every window edge
@pytest.mark.parametrize('shape', [(2, 60, 90, 512, 9), (1, 30, 45, 64, 9), (3, 17, 29, 96, 7), (3, 9, 82, 64, 9), (2, 23, 86, 32, 9), (1, 61, 88, 96, 9),
                                   (5, 60, 90, 64, 9)], ids=lambda s: 'B%d_%dx%d_%d-%d' % s)
def test_bf16_logits_layer_vs_bf16_oracle(shape):
    """The last (linear) layer on the bf16 path keeps its fp32 result: no output rounding, so the kernel must agree with
    the bf16-operand oracle to accumulation error (fp32 vs float64), orders of magnitude below the bf16-vs-fp32 gap."""
    from joint_cnn_mrf_amd.engine import Engine
    B, H, W, cin, cout = shape
    rs = np.random.RandomState(B * 1000 + cin)
    p = {'c/weights': (rs.standard_normal((9, 9, cin, cout)) * np.sqrt(2.0 / (81 * cin))).astype(np.float32),
         'c/biases': (0.1 * rs.standard_normal(cout)).astype(np.float32)}
    x = np.maximum(rs.standard_normal((B, H, W, cin)), 0).astype(np.float32)
    ref = O.conv_layer(x.astype(np.float64), p, 9, 1, 'c', last_layer=True, emulate='bf16')
    eng = Engine(device=0, precision='bf16').load_params(p)
    assert eng.conv_kernel_name('c', B, H, W) == ('conv_kxfold_bf16_kernel' if cout == 9 and 82 <= W <= 90 else 'conv_thin_bf16_kernel')
    got = eng.conv_layer(torch.as_tensor(x, device='cuda:0'), 'c', 1, last_layer=True, n_out=cout).cpu().numpy()
    eng.close()
    assert np.abs(got - ref).max() <= 2e-5 * np.abs(ref).max()
    assert np.abs(got - O.conv_layer(x.astype(np.float64), p, 9, 1, 'c', last_layer=True)).max() >= 1e-4 * np.abs(ref).max()   # ...which this test would not see


@pytest.mark.parametrize('hw', [(240, 368), (480, 720), (328, 488)])
def test_bf16_tower_other_resolutions_tracks_fp32(hw):
    """The bf16 path on image sizes other than 480x720 (odd map sizes exercise the patch / whole-row tile choices and the
    fused conv1+pool kernel's size conditions): logits stay within bf16 distance of the fp32 path."""
    from joint_cnn_mrf_amd import synth
    from joint_cnn_mrf_amd.engine import Engine
    p = synth.make_pd_params(debug=False, bn='trained')
    x = torch.as_tensor(synth.make_images(2, seed=9, height=hw[0], width=hw[1]), device='cuda:0')
    outs = {}
    for prec in ('fp32', 'bf16'):
        eng = Engine(device=0, precision=prec).load_params(p)
        outs[prec] = eng.model(x).cpu().numpy().astype(np.float64)
        eng.close()
    assert outs['fp32'].shape == outs['bf16'].shape
    scale = np.abs(outs['fp32']).max()
    assert np.abs(outs['bf16'] - outs['fp32']).max() <= 4e-2 * scale
    assert np.sqrt(np.mean((outs['bf16'] - outs['fp32']) ** 2)) <= 1e-2 * scale


@pytest.mark.parametrize('gain', [1e-9, 1.0, 3e7])
def test_split16_is_range_free(gain):
    """fp16x3 on inputs and weights far outside the fp16 range: every operand tensor is lifted by its own power-of-two
    scale (activations per launch, weights at pack time), so the kernel inherits fp32's range."""
    from joint_cnn_mrf_amd.engine import Engine
    rs = np.random.RandomState(77)
    p = layer_params(rs, 64, 128, 9)
    p['c/weights'] = (p['c/weights'] * np.float32(gain ** 0.5 if gain > 1 else 1.0)).astype(np.float32)
    x = (rs.standard_normal((3, 24, 64, 64)) * gain).astype(np.float32)
    ref = O.conv_layer(x.astype(np.float64), p, 9, 1, 'c')
    eng = Engine(device=0, f32_conv='split16', split_min_wgs=0).load_params(p)
    got = eng.conv_layer(torch.as_tensor(x, device='cuda:0'), 'c', 1, n_out=128).cpu().numpy()
    eng.close()
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() <= 2e-5 * np.abs(ref).max()


@pytest.mark.parametrize('case', ['constructive', 'tiny', 'huge'])
def test_bf16_fft_product_spectra_range(case):
    """The default bf16 route writes the product spectra Y as complex FP16 under a CONSTANT power-of-two shift (cgemm_split.hip, Y16): the shift
    comes from the bound |Y| < Cin 2^29.5 of the scaled operands, not from the data.  'constructive' drives that bound as hard as a layer can --
    every input 1, every weight the same positive number, so all Cin * 81 * H * W terms of the DC product add up in phase -- and must neither
    overflow (inf / NaN) nor lose the one-ulp class; 'tiny' / 'huge' inputs show that the per-image scale in front keeps the route range-free."""
    from joint_cnn_mrf_amd.engine import Engine
    rs = np.random.RandomState(5)
    B, H, W, cin, cout = 2, 30, 45, 512, 128
    if case == 'constructive':
        p = layer_params(rs, cin, cout, 9)
        p['c/weights'] = np.full((9, 9, cin, cout), 2.0 ** -12, np.float32)
        x = np.ones((B, H, W, cin), np.float32)
    else:
        p = layer_params(rs, cin, cout, 9)
        x = (np.maximum(rs.standard_normal((B, H, W, cin)), 0) * (1e-12 if case == 'tiny' else 1e12)).astype(np.float32)
    refb = O.conv_layer(x.astype(np.float64), p, 9, 1, 'c', emulate='bf16')
    eng = Engine(device=0, precision='bf16').load_params(p)
    assert eng.conv_kernel_name('c', B, H, W).startswith('conv_fft')
    got = eng.conv_layer(torch.as_tensor(x, device='cuda:0'), 'c', 1, n_out=cout).cpu().numpy().astype(np.float64)
    eng.close()
    assert np.isfinite(got).all()
    ulp = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(refb), 1e-300))) - 7)
    assert (np.abs(got - refb) <= 1.001 * ulp + 1e-3 * np.abs(refb).max()).all(), float(np.abs(got - refb).max() / np.abs(refb).max())


@pytest.mark.parametrize('geom', [(2, 60, 90, 30, 45, 15, 23, 128, 128), (1, 58, 90, 29, 45, 15, 23, 512, 64), (1, 32, 88, 16, 44, 8, 22, 128, 128),
                                  (1, 24, 40, 24, 40, 6, 10, 128, 128)], ids=lambda g: 'B%d_%dx%d_%dx%d_%dx%d_%d-%d' % g)
def test_merged_conv_layer_vs_oracle(geom):
    """x = (x1 + up(x2) + up(x3)) / 3 followed by conv5 (main.py:58,67,69-71) through jcm_conv_layer_merged: on the frequency-domain route the
    merge is formed inside the layer's forward row pass.  60x90 / 30x45 / 15x23 is the model's geometry (bf16 handles: the register kernel with
    compile-time taps, rows_fwd_merge_reg_kernel -- with the tower's 512 channels its work groups are laid out per XCD, and 58 rows leave the last group of
    four rows incomplete); the others take the generic kernel (one with x2 already at full size)."""
    from joint_cnn_mrf_amd.engine import Engine
    B, H, W, H2, W2, H3, W3, cin, cout = geom
    rs = np.random.RandomState(H * 100 + W)
    p = layer_params(rs, cin, cout, 9)
    xs = [np.maximum(rs.standard_normal((B, h, w, cin)), 0).astype(np.float32) for h, w in ((H, W), (H2, W2), (H3, W3))]
    up = lambda t: O.resize_bilinear_tf1(t, H, W)
    merged = (xs[0].astype(np.float64) + up(xs[1].astype(np.float64)) + up(xs[2].astype(np.float64))) / 3.0
    ref = O.conv_layer(merged, p, 9, 1, 'c')
    xd = [torch.as_tensor(t, device='cuda:0') for t in xs]
    eng = Engine(device=0).load_params(p)
    assert eng.conv_kernel_name('c', B, H, W).startswith('conv_fft')
    got = eng.conv_layer_merged(xd[0], xd[1], xd[2], 'c', cout).cpu().numpy()
    eng.close()
    assert np.abs(got - ref).max() <= 2e-5 * np.abs(ref).max()
    # bf16 handle against the oracle in its arithmetic: bf16 maps in, the merged map rounded to bf16 (oracle.model), the layer on bf16 operands
    xb = [O.bf16_round(t.astype(np.float64)) for t in xs]
    mb = O.bf16_round((xb[0] + up(xb[1]) + up(xb[2])) / 3.0)
    refb = O.conv_layer(mb, p, 9, 1, 'c', emulate='bf16')
    eng = Engine(device=0, precision='bf16').load_params(p)
    assert eng.conv_kernel_name('c', B, H, W).startswith('conv_fft')
    gotb = eng.conv_layer_merged(xd[0], xd[1], xd[2], 'c', cout).cpu().numpy().astype(np.float64)
    eng.close()
    check_bf16_layer(gotb, refb, slack_rel=1e-3, flips=0.12, rms_rel=4e-4)
    eng = Engine(device=0, precision='bf16', fft_single=False, fft_t16=False).load_params(p)      # the strict arm: generic merge kernel, fp32 T
    gotb = eng.conv_layer_merged(xd[0], xd[1], xd[2], 'c', cout).cpu().numpy().astype(np.float64)
    eng.close()
    # (a merged value that fp32 and float64 round to different bf16 numbers -- a few per million -- moves the 81 x 128 outputs under it by 2^-8 of one
    # term: the slack of the one-ulp bar is 2e-4 of the scale here instead of the plain layer's 1e-5)
    check_bf16_layer(gotb, refb, slack_rel=2e-4, flips=0.03)


@pytest.mark.parametrize('seed', [101, 202, 303])
def test_tower_more_seeds_debug_width(seed):
    """The whole tower (part detector + spatial model + arg-max) on fresh seeded weights / images / priors at --debug width,
    in the three fp32 convolution modes: heat maps within 1e-4, arg-max identical wherever the oracle's own top-2 margin
    is not at rounding level."""
    from joint_cnn_mrf_amd import synth
    from joint_cnn_mrf_amd.engine import Engine
    p = synth.make_pd_params(debug=True, seed=seed, bn='trained', conv6_gain=6.0)
    p.update(synth.make_sm_params(synth.synthetic_priors(seed=seed + 1), kind='trained', seed=seed + 2))
    x, torso = synth.make_images(2, seed=seed + 3), synth.make_torso(2, seed=seed + 4)
    ref = O.forward(x.astype(np.float64), torso.astype(np.float64), p)
    for mode in ('exact', 'split16'):
        eng = Engine(device=0, f32_conv=mode, split_min_wgs=0).load_params(p)
        r = eng.forward(torch.as_tensor(x, device='cuda:0'), torch.as_tensor(torso, device='cuda:0'), use_sm=True)
        eng.close()
        for k in ('pd_prob', 'sm_prob'):
            got = r[k].cpu().numpy()
            assert np.abs(got - ref[k]).max() <= 1e-4, (mode, k)
            flat = ref[k].reshape(2, 5400, 9)
            top2 = np.sort(flat, axis=1)[:, -2:, :]
            safe = (top2[:, 1, :] - top2[:, 0, :]) > 1e-6 * top2[:, 1, :]
            ck = 'pd_coords' if k == 'pd_prob' else 'sm_coords'
            same = (r[ck].cpu().numpy() == ref[ck]).all(axis=1)
            assert (same | ~safe).all(), (mode, ck)


def test_filter_spectra_cache_is_bounded():
    """conv_fft keeps the filter spectra per (layer, map size); with the bound at 0 GB every new (layer, size) evicts the others, and the
    results stay those of the oracle (a process of its own: the bound is read once)."""
    import os
    import subprocess
    import sys
    code = r"""
import numpy as np, torch
import joint_cnn_mrf_amd
from joint_cnn_mrf_amd.engine import Engine
from oracle import jcm_oracle as O
rs = np.random.RandomState(5)
p = {}
for s, (cin, cout) in {'a': (64, 64), 'b': (128, 64)}.items():
    p[s + '/weights'] = (rs.standard_normal((9, 9, cin, cout)) * np.sqrt(2.0 / (81 * cin))).astype(np.float32)
    p[s + '/biases'] = (0.1 * rs.standard_normal(cout)).astype(np.float32)
eng = Engine(device=0).load_params(p)
for rep in range(2):
    for s, cin, (H, W) in (('a', 64, (20, 30)), ('b', 128, (12, 40)), ('a', 64, (33, 17)), ('b', 128, (12, 40))):
        x = rs.standard_normal((2, H, W, cin)).astype(np.float32)
        assert eng.conv_kernel_name(s, 2, H, W).startswith('conv_fft')
        got = eng.conv_layer(torch.as_tensor(x, device='cuda:0'), s, 1, last_layer=True, n_out=64).cpu().numpy()
        ref = O.conv_layer(x.astype(np.float64), p, 9, 1, s, last_layer=True)
        assert np.abs(got - ref).max() <= 2e-5 * np.abs(ref).max()
print('ok')
"""
    env = dict(os.environ, JCM_FFT_CACHE_GB='0', PYTHONPATH=os.pathsep.join([os.path.dirname(os.path.dirname(os.path.abspath(__file__)))] + sys.path))
    r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'ok' in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
