"""Same-box A/B of one integer option of a bf16 (default) or fp32 engine: part-detector logits at both settings against each other
(bit-identical or not, max / rms of the difference over the logit scale), then the time of the full forward at each setting, interleaved.
    python tools/ab_option.py OPTION V0 V1 [B=256] [precision=bf16]"""
import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import joint_cnn_mrf_amd
from joint_cnn_mrf_amd import synth
from joint_cnn_mrf_amd.engine import Engine

OPT, V0, V1 = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
BT = int(sys.argv[4]) if len(sys.argv) > 4 else 256
PREC = sys.argv[5] if len(sys.argv) > 5 else 'bf16'
p = synth.make_pd_params(debug=False, bn='trained', conv6_gain=8.0)
p.update(synth.make_sm_params(synth.synthetic_priors(), kind='trained'))
x = torch.as_tensor(synth.make_images(8, seed=5), device='cuda:0')
eng = Engine(device=0, precision=PREC).load_params(p)
outs = {}
for v in (V0, V1):
    eng.set_option(OPT, v)
    outs[v] = eng.model(x).float().cpu().numpy()
a, b = outs[V0], outs[V1]
scale = np.abs(a).max()
print('%s=%d vs %d: bit-identical %s, max |d| / scale %.3e, rms / scale %.3e' % (OPT, V0, V1, np.array_equal(a, b), np.abs(a - b).max() / scale, np.sqrt(np.mean((a - b) ** 2)) / scale))
xt = torch.as_tensor(synth.make_images(BT, seed=7), device='cuda:0')
tt = torch.as_tensor(synth.make_torso(BT, seed=8), device='cuda:0')
for rep in range(3):
    for v in (V0, V1):
        eng.set_option(OPT, v)
        for _ in range(2):
            eng.forward(xt, tt, want_prob=False)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(9)]
        for i in range(8):
            ev[i].record(); eng.forward(xt, tt, want_prob=False)
        ev[8].record(); torch.cuda.synchronize()
        ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(8))
        print('B %d %s=%d: median %.3f ms, min %.3f ms' % (BT, OPT, v, ms[4], ms[0]), flush=True)
eng.close()
