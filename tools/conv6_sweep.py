"""Logits-layer launch time against Cin at batch 256, 60x90 (per-chunk slope and per-tile intercept of conv_kxfold_bf16).
Run on the GPU box; JCM_KXFOLD=0 selects conv_thin_bf16."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from joint_cnn_mrf_amd.engine import Engine  # noqa: E402

B = int(os.environ.get('B', '256'))
for cin in (32, 128, 256, 512):
    rs = np.random.RandomState(cin)
    p = {'c/weights': (rs.standard_normal((9, 9, cin, 9)) * np.sqrt(2.0 / (81 * cin))).astype(np.float32),
         'c/biases': (0.1 * rs.standard_normal(9)).astype(np.float32)}
    eng = Engine(device=0, precision='bf16').load_params(p)
    x = torch.rand((B, 60, 90, cin), device='cuda:0')
    for _ in range(2):
        eng.conv_layer(x, 'c', 1, last_layer=True, n_out=9)
    eng.set_profile(True)
    for _ in range(5):
        eng.conv_layer(x, 'c', 1, last_layer=True, n_out=9)
    torch.cuda.synchronize()
    eng.set_profile(False)
    ms, n = eng.profile_read('c')
    print('Cin %4d  %s  %.3f ms' % (cin, eng.conv_kernel_name('c', B, 60, 90), ms / n))
    eng.close()
