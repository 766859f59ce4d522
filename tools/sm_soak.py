"""Two engines, two streams, the call chain off: the spatial model's balanced kernel (work groups that wait for a partial sum of the previous slot of
their XCD) running beside ITSELF.  Every result must equal the engine's own single-stream result, and the run must end (no wait cycle).
    python tools/sm_soak.py [iterations=200] [B=37]"""
import sys, threading
import numpy as np, torch
sys.path.insert(0, '.')
import joint_cnn_mrf_amd  # noqa: F401
from joint_cnn_mrf_amd import synth
from joint_cnn_mrf_amd.engine import Engine

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
B = int(sys.argv[2]) if len(sys.argv) > 2 else 37
p = synth.make_pd_params(debug=True, bn='trained', conv6_gain=8.0)
p.update(synth.make_sm_params(synth.synthetic_priors(), kind='trained'))
streams = [torch.cuda.Stream(device=0) for _ in range(2)]
engs, hms, refs = [], [], []
for i, st in enumerate(streams):
    with torch.cuda.stream(st):
        e = Engine(device=0, stream=st, call_order=False).load_params(p)
        hm = torch.rand((B, 60, 90, 10), device='cuda:0', generator=torch.Generator(device='cuda:0').manual_seed(i)) ** 8 * 0.02
        engs.append(e); hms.append(hm); refs.append(e.spatial_model(hm).clone())
torch.cuda.synchronize()
bad = [0, 0]


def work(i):
    torch.cuda.set_device(0)
    with torch.cuda.stream(streams[i]):
        for _ in range(iters):
            out = engs[i].spatial_model(hms[i])
            if not torch.equal(out, refs[i]):
                bad[i] += 1
        streams[i].synchronize()


th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
[t.start() for t in th]
[t.join() for t in th]
print('B', B, 'iterations', iters, 'differing results', bad)
assert bad == [0, 0]
for e in engs:
    e.close()
