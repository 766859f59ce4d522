"""How many torch threads should the CPU baseline of bench.py use?  (run on the GPU box: python tools/cpu_threads.py)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import joint_cnn_mrf_amd  # noqa: F401
from joint_cnn_mrf_amd import synth
from oracle import jcm_oracle_torch as T
p = synth.make_pd_params(debug=False)
p.update(synth.make_sm_params(synth.synthetic_priors(), kind='init'))
x, torso = synth.make_images(8, seed=99), synth.make_torso(8, seed=98)
T.forward(x[:1], torso[:1], p, dtype=torch.float32)
for n in (128, 64, 32):
    torch.set_num_threads(n)
    for b in (1, 8):
        t0 = time.time(); T.forward(x[:b], torso[:b], p, dtype=torch.float32); dt = time.time() - t0
        print('%d threads, B=%d: %.2f s = %.3f images/s' % (n, b, dt, b / dt), flush=True)
