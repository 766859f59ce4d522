"""Stability soak of the training step: N Adam updates on a small synthetic data set in every arithmetic mode;
prints the loss trajectory (tools script, not part of the test suite).   python tools/train_soak.py [steps]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
import joint_cnn_mrf_amd  # noqa: F401
from joint_cnn_mrf_amd import synth
from joint_cnn_mrf_amd.engine import Engine
from joint_cnn_mrf_amd.train import Trainer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 120
p = synth.make_pd_params(debug=False)
p.update(synth.make_sm_params(synth.synthetic_priors(), kind='init'))
data = [(torch.as_tensor(synth.make_images(8, seed=300 + i), device='cuda:0'), torch.as_tensor(synth.make_targets(8, seed=400 + i), device='cuda:0')) for i in range(4)]
# 'exact': the default fp32 engine (frequency domain, two scaled fp16 parts); 'chain': the fp32 MFMA kernels; 'split16': the direct fp16x3 kernels
for mode in ('exact', 'chain', 'split16', 'bf16'):
    kw = (dict(precision='bf16') if mode == 'bf16' else dict(conv9_fft=False) if mode == 'chain'
          else dict(f32_conv=mode))
    eng = Engine(device=0, **kw).load_params(p)
    tr = Trainer(eng, optimizer='adam', lr=0.001, lmbd=0.001, use_sm=True, n_updates_total=steps)
    hist, t0 = [], time.time()
    for s in range(steps):
        x, y = data[s % len(data)]
        losses, norm = tr.train_step(x, y, want_norm=(s % 20 == 0))
        if s % 20 == 0 or s == steps - 1:
            l = losses.cpu().numpy()
            hist.append((s, float(l[1]), float(l[2]), norm))
    torch.cuda.synchronize()
    print('%-8s %5.1f s  ' % (mode, time.time() - t0) + '  '.join('s%d pd %.3f sm %.3f' % (s, a, b) for s, a, b, _ in hist))
    assert all(np.isfinite([a, b]).all() for _, a, b, _ in hist)
    eng.close()
