"""Times jcm_sm_forward (the spatial model alone) with HIP events: python tools/sm_time.py [B ...]"""
import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import joint_cnn_mrf_amd
from joint_cnn_mrf_amd import synth
from joint_cnn_mrf_amd.engine import Engine
p = synth.make_pd_params(debug=True, bn='trained', conv6_gain=8.0)
p.update(synth.make_sm_params(synth.synthetic_priors(), kind='trained'))
eng = Engine(device=0).load_params(p)
for B in [int(a) for a in sys.argv[1:]] or [64, 256]:
    hm = torch.rand((B, 60, 90, 10), device='cuda:0') * 1e-3
    for _ in range(3):
        out = eng.spatial_model(hm)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(11)]
    for i in range(10):
        ev[i].record(); out = eng.spatial_model(hm)
    ev[10].record(); torch.cuda.synchronize()
    ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(10))
    print('B', B, 'spatial model median ms', round(ms[5], 4), 'min', round(ms[0], 4), flush=True)
eng.close()
