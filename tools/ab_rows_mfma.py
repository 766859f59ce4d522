"""Same-box A/B of option "fft_rows_mfma" (bf16 handles: 96-point row passes on the matrix cores, conv_fft_rows_mfma.hip):
logits of the part detector with the option at 0 and at its default on the same images, against each other and against the
fp32 engine, then the time of the bf16 forward at each setting (interleaved).  python tools/ab_rows_mfma.py [B_time] [value=1]"""
import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import joint_cnn_mrf_amd
from joint_cnn_mrf_amd import synth
from joint_cnn_mrf_amd.engine import Engine

BT = int(sys.argv[1]) if len(sys.argv) > 1 else 256
BITS = int(sys.argv[2]) if len(sys.argv) > 2 else 1
p = synth.make_pd_params(debug=False, bn='trained', conv6_gain=8.0)
p.update(synth.make_sm_params(synth.synthetic_priors(), kind='trained'))
x = torch.as_tensor(synth.make_images(8, seed=5), device='cuda:0')
ref = Engine(device=0, precision='fp32').load_params(p)
lr = ref.model(x).float().cpu().numpy()
ref.close()
eng = Engine(device=0, precision='bf16').load_params(p)
outs = {}
for v in (0, BITS):
    eng.set_option('fft_rows_mfma', v)
    outs[v] = eng.model(x).float().cpu().numpy()
scale = np.abs(lr).max()
a, b = outs[0], outs[BITS]
print('logit scale %.4g' % scale)
print('mfma vs register kernels: max |d| / scale %.3e, rms / scale %.3e, differing values %.4f' % (np.abs(a - b).max() / scale, np.sqrt(np.mean((a - b) ** 2)) / scale, np.mean(a != b)))
for v in (0, BITS):
    d = outs[v] - lr
    am = np.mean(outs[v].reshape(8, -1, outs[v].shape[-1]).argmax(1) == lr.reshape(8, -1, lr.shape[-1]).argmax(1))
    print('fft_rows_mfma=%d vs fp32 engine: max |d| / scale %.3e, rms / scale %.3e, arg-max equal %.3f' % (v, np.abs(d).max() / scale, np.sqrt(np.mean(d ** 2)) / scale, am))
xt = torch.as_tensor(synth.make_images(BT, seed=7), device='cuda:0')
tt = torch.as_tensor(synth.make_torso(BT, seed=8), device='cuda:0')
for rep in range(3):
    for v in (0, BITS):
        eng.set_option('fft_rows_mfma', v)
        for _ in range(2):
            eng.forward(xt, tt, want_prob=False)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(7)]
        for i in range(6):
            ev[i].record(); eng.forward(xt, tt, want_prob=False)
        ev[6].record(); torch.cuda.synchronize()
        ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(6))
        print('B %d fft_rows_mfma=%d: median %.3f ms, min %.3f ms' % (BT, v, ms[3], ms[0]), flush=True)
eng.close()
