import sys, time, ctypes, torch, numpy as np
sys.path.insert(0, '.')
import joint_cnn_mrf_amd
from joint_cnn_mrf_amd import synth, _lib
from joint_cnn_mrf_amd.engine import Engine
p = synth.make_sm_params(synth.synthetic_priors(), kind='init')
eng = Engine(device=0).load_params(p)
B = 256
hm = torch.rand(B, 60, 90, 10, device='cuda:0')
for algo in ('fft', 'fft_split'):
    eng.set_sm_algo(algo)
    for chunk in (64, 32, 16, 8, 4):
        _lib.check(eng._lib.jcm_set_option(eng._h, b'sm_chunk', chunk), 'opt')
        for _ in range(2): eng.spatial_model(hm)
        torch.cuda.synchronize(); t = time.time()
        for _ in range(5): out = eng.spatial_model(hm)
        torch.cuda.synchronize(); print('%-9s sm_chunk %3d: %.3f ms per B=256 forward' % (algo, chunk, (time.time() - t) / 5 * 1e3))
