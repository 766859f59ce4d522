"""Is a forward bit-reproducible?  One engine / one stream, then two engines on two streams; per stage (part detector logits, spatial
model) so that a difference can be placed.  Run on the GPU box: python tools/determinism.py [fp32|bf16]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import joint_cnn_mrf_amd  # noqa: E402,F401
from golden_util import flic_priors, full_inputs, seeds  # noqa: E402
from joint_cnn_mrf_amd import synth  # noqa: E402
from joint_cnn_mrf_amd.engine import Engine  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else 'fp32'
FFT = not (len(sys.argv) > 2 and sys.argv[2] == 'nofft')
x, torso, p = full_inputs()
p.update(synth.make_sm_params(flic_priors(), kind='trained', seed=seeds()['sm']))
xd = torch.as_tensor(x, device='cuda:0')
td = torch.as_tensor(torso, device='cuda:0')


def diff(a, b):
    a, b = a.cpu().numpy(), b.cpu().numpy()
    return int((a != b).sum()), float(np.abs(a.astype(np.float64) - b).max())


eng = Engine(device=0, precision=prec, conv9_fft=FFT).load_params(p)
logits = [eng.model(xd) for _ in range(6)]
torch.cuda.synchronize()
print('one engine, model():', [diff(logits[0], l) for l in logits[1:]])
fw = [eng.forward(xd, td, use_sm=True) for _ in range(6)]
torch.cuda.synchronize()
for k in ('pd_prob', 'sm_prob', 'sm_coords'):
    print('one engine, forward()', k, [diff(fw[0][k], f[k]) for f in fw[1:]])
eng.close()

SAME = len(sys.argv) > 3 and sys.argv[3] == 'same'
streams = [torch.cuda.current_stream()] * 2 if SAME else [torch.cuda.Stream(device='cuda:0') for _ in range(2)]
engs = [Engine(device=0, precision=prec, stream=s, conv9_fft=FFT).load_params(p) for s in streams]
torch.cuda.synchronize()
outs = [[], []]
for _ in range(int(os.environ.get("DET_ITERS", "20"))):
    for e, (en, s) in enumerate(zip(engs, streams)):
        with torch.cuda.stream(s):
            outs[e].append(en.forward(xd, td, use_sm=True))
torch.cuda.synchronize()
for e in range(2):
    for k in ('pd_prob', 'sm_prob'):
        print('two engines: engine', e, k, [diff(outs[e][0][k], o[k])[0] for o in outs[e][1:]])
print('engine 0 vs engine 1:', diff(outs[0][0]['sm_prob'], outs[1][0]['sm_prob']))

# ---- which layer?  each engine repeats one layer on its own stream, both at the same time
shapes = {'conv1_fullres': (480, 720, 3, 64, 2), 'conv2_fullres': (120, 180, 64, 128, 1), 'conv3_fullres': (60, 90, 128, 256, 1), 'conv4_fullres': (60, 90, 256, 512, 1),
          'conv5': (60, 90, 512, 512, 1), 'conv6': (60, 90, 512, 9, 1), 'conv4_quarterres': (15, 23, 256, 512, 1)}
for scope, (H, W, cin, cout, stride) in shapes.items():
    xi = torch.rand((2, H, W, cin), device='cuda:0')
    torch.cuda.synchronize()
    res = [[], []]
    for _ in range(12):
        for e, (en, s) in enumerate(zip(engs, streams)):
            with torch.cuda.stream(s):
                try:
                    res[e].append(en.conv_layer(xi, scope, stride, n_out=cout, last_layer=(scope == 'conv6')))
                except Exception as ex:      # a layer without a stand-alone kernel on this handle
                    res[e].append(None)
                    err = ex
    torch.cuda.synchronize()
    if res[0][0] is None:
        print('  layer', scope, 'skipped:', err)
        continue
    print('  layer %-18s' % scope, [[diff(r[0], o)[0] for o in r[1:]] for r in res], 'engine0 vs engine1', diff(res[0][0], res[1][0])[0])

# ---- where do the differences sit?  (conv4_fullres, first differing run of engine 0)
H, W, cin, cout, stride = shapes['conv4_fullres']
xi = torch.rand((2, H, W, cin), device='cuda:0')
torch.cuda.synchronize()
res = [[], []]
for _ in range(12):
    for e, (en, s) in enumerate(zip(engs, streams)):
        with torch.cuda.stream(s):
            res[e].append(en.conv_layer(xi, 'conv4_fullres', stride, n_out=cout))
torch.cuda.synchronize()
ref = engs[0].conv_layer(xi, 'conv4_fullres', stride, n_out=cout)
torch.cuda.synchronize()
ref = ref.cpu().numpy()
for e in range(2):
    for i, o in enumerate(res[e]):
        d = o.cpu().numpy() != ref
        if d.any():
            idx = np.argwhere(d)
            print('engine', e, 'run', i, 'n', len(idx), 'images', np.unique(idx[:, 0]), 'rows', np.unique(idx[:, 1])[:12], 'cols', np.unique(idx[:, 2])[:12],
                  'channels', np.unique(idx[:, 3])[:16], '... n_ch', len(np.unique(idx[:, 3])), 'max|d|', float(np.abs(o.cpu().numpy() - ref).max()))
            break

# ---- are stores dropped?  outputs pre-filled with NaN on the engine's stream
for en in engs:
    orig = en._new
    en._new = (lambda o: (lambda *a, **k: o(*a, **k).fill_(float('nan')) if k.get('dtype', torch.float32) == torch.float32 else o(*a, **k)))(orig)
res = [[], []]
for _ in range(12):
    for e, (en, s) in enumerate(zip(engs, streams)):
        with torch.cuda.stream(s):
            res[e].append(en.conv_layer(xi, 'conv4_fullres', stride, n_out=cout))
torch.cuda.synchronize()
for e in range(2):
    for i, o in enumerate(res[e]):
        o = o.cpu().numpy()
        d = o != ref
        if d.any():
            print('NaN-prefilled: engine', e, 'run', i, 'differing', int(d.sum()), 'of which NaN', int(np.isnan(o).sum()))
