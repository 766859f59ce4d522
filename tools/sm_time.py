"""Spatial model alone at batch 256 (wall clock over 5 forwards per algorithm).  Run on the GPU box."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from joint_cnn_mrf_amd import synth  # noqa: E402
from joint_cnn_mrf_amd.engine import Engine  # noqa: E402

p = synth.make_sm_params(synth.synthetic_priors(), kind='trained')
eng = Engine(device=0).load_params(p)
B = int(os.environ.get('B', '256'))
hm = torch.rand(B, 60, 90, 10, device='cuda:0')
outs = {}
for algo in ('fft_split', 'fft_fused'):
    eng.set_sm_algo(algo)
    for _ in range(2):
        eng.spatial_model(hm)
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(5):
        outs[algo] = eng.spatial_model(hm)
    torch.cuda.synchronize()
    print('%-9s %.3f ms per B=%d forward' % (algo, (time.time() - t) / 5 * 1e3, B))
d = (outs['fft_split'] - outs['fft_fused']).abs().max().item()
print('max |fft_split - fft_fused| = %.3g (logit scale %.3g)' % (d, outs['fft_split'].abs().max().item()))
eng.close()
