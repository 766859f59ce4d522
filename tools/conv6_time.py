"""Time the logits layer (conv6) inside the bf16 tower at batch 256: per-launch HIP events of the library's own profile
option.  JCM_KXFOLD=0 selects conv_thin_bf16_kernel (the A/B arm).  Run on the GPU box:
    python tools/conv6_time.py; JCM_KXFOLD=0 python tools/conv6_time.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from joint_cnn_mrf_amd import synth  # noqa: E402
from joint_cnn_mrf_amd.engine import Engine  # noqa: E402

B = int(os.environ.get('B', '256'))
p = synth.make_pd_params(debug=False, bn='trained')
p.update(synth.make_sm_params(synth.synthetic_priors(), kind='trained'))
eng = Engine(device=0, precision='bf16').load_params(p)
x = torch.as_tensor(synth.make_images(B, seed=1), device='cuda:0')
t = torch.as_tensor(synth.make_torso(B, seed=2), device='cuda:0')
for _ in range(2):
    eng.forward(x, t, use_sm=True, want_prob=False)
eng.set_profile(True)
for _ in range(5):
    eng.forward(x, t, use_sm=True, want_prob=False)
torch.cuda.synchronize()
eng.set_profile(False)
print('kernel', eng.conv_kernel_name('conv6', B, 60, 90))
for scope in os.environ.get('SCOPES', 'conv4_fullres conv5 conv6 conv4_quarterres').split():
    ms, n = eng.profile_read(scope)
    print('%-18s %.3f ms x %d' % (scope, ms / max(n, 1), n))
eng.close()
