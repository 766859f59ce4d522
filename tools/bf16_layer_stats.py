import sys, numpy as np, torch
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import joint_cnn_mrf_amd
from golden_util import full_inputs
from oracle import jcm_oracle as O
from joint_cnn_mrf_amd.engine import Engine
x, torso, p = full_inputs()
from joint_cnn_mrf_amd import synth
p.update(synth.make_sm_params(synth.synthetic_priors(), kind='trained'))
taps = {}
ref = O.model(x.astype(np.float64), p, emulate='bf16', taps=taps)
dev = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32), device='cuda:0')
for single, t16 in ((True, True), (True, False), (False, False)):
    eng = Engine(device=0, precision='bf16', fft_single=single, fft_t16=t16).load_params(p)
    logits = eng.model(dev(x)).cpu().numpy().astype(np.float64)
    scale = np.abs(ref).max(); err = np.abs(logits - ref)
    print('fft_single=%s fft_t16=%s tower: max %.2e rms %.2e of scale' % (single, t16, err.max() / scale, np.sqrt((err ** 2).mean()) / scale))
    for scope, tin, tout in (('conv5', 'merge', 'conv5'), ('conv4_halfres', 'conv3_halfres', 'conv4_halfres'), ('conv4_fullres', 'conv3_fullres', 'conv4_fullres'), ('conv4_quarterres', 'conv3_quarterres', 'conv4_quarterres')):
        got = eng.conv_layer(dev(taps[tin]), scope, 1, n_out=512).cpu().numpy().astype(np.float64)
        r = taps[tout]
        ulp = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(r), 1e-30))) - 7)
        d = np.abs(got - r); sc = np.abs(r).max()
        over = np.maximum(d - 1.001 * ulp, 0)
        print('   %-18s flips %.2f %%  rms/scale %.2e  max excess over one ulp / scale %.2e   entries beyond 1 ulp %.4f %%' % (scope, 100 * (d > 0).mean(), np.sqrt((d ** 2).mean()) / sc, over.max() / sc, 100 * (over > 0).mean()))
    if '--time' in sys.argv:      # same-box A/B of the whole forward at configs[2]'s batch
        xb, tb = dev(synth.make_images(256)), dev(synth.make_torso(256))
        for _ in range(3): eng.forward(xb, tb, use_sm=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): eng.forward(xb, tb, use_sm=True)
        e1.record(); torch.cuda.synchronize()
        print('   forward of 256 images: %.2f ms' % (e0.elapsed_time(e1) / 10))
        del xb, tb
    eng.close()
