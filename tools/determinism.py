"""Is a forward bit-reproducible?  One engine / one stream, then two engines on two streams (one host thread alternating, or one host
thread per engine); per stage (part detector, spatial model), per layer, and per PAIR of layers (engine 0 repeats layer X while engine 1
repeats layer Y) so that a difference can be placed.  Run on the GPU box:

    python tools/determinism.py [fp32|bf16] [nofft] [same]

Environment: DET_ITERS (alternating forwards per engine, default 20), DET_LAYER_ITERS (per-layer repeats, default 12),
DET_CHAIN=0 takes both engines out of the per-device call chain (jcm_set_option "call_order" 0: what would the chain hide?),
DET_THREADS=1 drives each engine from its own host thread, DET_PAIRS=1 adds the layer-pair matrix, DET_BATCH (images, default 2),
DET_FP16=0 (fp32 engines: three bf16 parts instead of two scaled fp16 parts)."""
import os
import sys
import threading

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
SAME = len(sys.argv) > 3 and sys.argv[3] == 'same'
ITERS = int(os.environ.get('DET_ITERS', '20'))
LITERS = int(os.environ.get('DET_LAYER_ITERS', '12'))
CHAIN = os.environ.get('DET_CHAIN', '1') != '0'
THREADS = os.environ.get('DET_THREADS', '0') == '1'
PAIRS = os.environ.get('DET_PAIRS', '0') == '1'
NB = int(os.environ.get('DET_BATCH', '2'))
FP16 = os.environ.get('DET_FP16', '1') != '0'      # fp32 engines: 0 = three bf16 parts instead of two scaled fp16 parts (no scale words)
x, torso, p = full_inputs()
p.update(synth.make_sm_params(flic_priors(), kind='trained', seed=seeds()['sm']))
if NB != 2:
    x, torso = synth.make_images(NB, seed=91), synth.make_torso(NB, seed=92)
xd = torch.as_tensor(x, device='cuda:0')
td = torch.as_tensor(torso, device='cuda:0')
print('precision %s fft %s chain %s threads %s iters %d batch %d' % (prec, FFT, CHAIN, THREADS, ITERS, NB), flush=True)


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
ref_fw = {k: fw[0][k].clone() for k in ('pd_prob', 'sm_prob')}
eng.close()

streams = [torch.cuda.current_stream()] * 2 if SAME else [torch.cuda.Stream(device='cuda:0') for _ in range(2)]
engs = [Engine(device=0, precision=prec, stream=s, conv9_fft=FFT, call_order=CHAIN).load_params(p) for s in streams]
torch.cuda.synchronize()


def run_both(fn, iters):
    """fn(engine index) -> result, `iters` times per engine, alternating from this thread or from one thread per engine"""
    outs = [[], []]

    def drive(e):
        with torch.cuda.stream(streams[e]):
            for _ in range(iters):
                outs[e].append(fn(e))

    if THREADS:
        ts = [threading.Thread(target=drive, args=(e,)) for e in range(2)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
    else:
        for _ in range(iters):
            for e in range(2):
                with torch.cuda.stream(streams[e]):
                    outs[e].append(fn(e))
    torch.cuda.synchronize()
    return outs


# results are compared on the device (a mismatch count per run), so thousands of iterations need no host copies
def count_bad(outs, ref, key=None):
    bad = []
    for i, o in enumerate(outs):
        t = o[key] if key else o
        if not torch.equal(t, ref):
            bad.append(i)
    return bad


CH = 50
tot = {(e, k): [] for e in range(2) for k in ('pd_prob', 'sm_prob')}
for c0 in range(0, ITERS, CH):
    outs = run_both(lambda e: engs[e].forward(xd, td, use_sm=True), min(CH, ITERS - c0))
    for e in range(2):
        for k in ('pd_prob', 'sm_prob'):
            tot[(e, k)] += [c0 + i for i in count_bad(outs[e], ref_fw[k], k)]
    del outs
for (e, k), bad in sorted(tot.items()):
    print('two engines: engine', e, k, 'mismatching forwards %d of %d' % (len(bad), ITERS), bad[:20], flush=True)

# ---- which layer?  each engine repeats one layer on its own stream, both at the same time
shapes = {'conv1_fullres': (480, 720, 3, 64, 2), 'conv2_fullres': (120, 180, 64, 128, 1), 'conv3_fullres': (60, 90, 128, 256, 1), 'conv4_fullres': (60, 90, 256, 512, 1),
          'conv5': (60, 90, 512, 512, 1), 'conv6': (60, 90, 512, 9, 1), 'conv4_halfres': (30, 45, 256, 512, 1), 'conv4_quarterres': (15, 23, 256, 512, 1)}
ONLY = os.environ.get('DET_ONLY')      # comma-separated layer scopes: restrict the per-layer section
if ONLY:
    shapes = {k: v for k, v in shapes.items() if k in ONLY.split(',')}
inputs = {s: torch.rand((NB, v[0], v[1], v[2]), device='cuda:0') for s, v in shapes.items()}
torch.cuda.synchronize()


shapes.update({'conv2_halfres': (60, 90, 64, 128, 1), 'conv3_halfres': (30, 45, 128, 256, 1), 'conv2_quarterres': (30, 45, 64, 128, 1), 'conv3_quarterres': (15, 23, 128, 256, 1)})
for s_ in ('conv2_halfres', 'conv3_halfres', 'conv2_quarterres', 'conv3_quarterres'):
    inputs[s_] = torch.rand((NB, shapes[s_][0], shapes[s_][1], shapes[s_][2]), device='cuda:0')
shapes['sm'] = (60, 90, 10, 9, 1)          # pseudo-layers: the spatial model, and the whole part detector
shapes['model'] = (480, 720, 3, 9, 1)
if ONLY:
    shapes = {k: v for k, v in shapes.items() if k in ONLY.split(',')}
for s_ in ('sm', 'model'):
    if s_ in shapes:
        inputs[s_] = torch.rand((NB, shapes[s_][0], shapes[s_][1], shapes[s_][2]), device='cuda:0')
torch.cuda.synchronize()


def layer_call(e, scope):
    if scope == 'sm':
        return engs[e].spatial_model(inputs[scope])
    if scope == 'model':
        return engs[e].model(inputs[scope])
    H, W, cin, cout, stride = shapes[scope]
    return engs[e].conv_layer(inputs[scope], scope, stride, n_out=cout, last_layer=(scope == 'conv6'))


refs = {}
for scope in shapes:
    try:
        with torch.cuda.stream(streams[0]):      # (the engine launches on ITS stream: the clone must queue behind it)
            refs[scope] = layer_call(0, scope).clone()
        torch.cuda.synchronize()
    except Exception as ex:      # a layer without a stand-alone kernel on this handle
        print('  layer', scope, 'skipped:', ex)
for scope in refs:
    res = run_both(lambda e: layer_call(e, scope), LITERS)
    print('  layer %-18s mismatching runs (engine 0, engine 1) of %d:' % (scope, LITERS), [len(count_bad(r, refs[scope])) for r in res], flush=True)
    del res
# ---- what property of the co-resident work matters?  engine 0 repeats conv6 while stream 1 runs a torch-only aggressor (no libjcm kernel)
AGGR = os.environ.get('DET_AGGR')
if AGGR and 'conv6' in refs:
    big = torch.rand(64 << 20, device='cuda:0')                   # 256 MB
    ma = torch.randn(8192, 8192, device='cuda:0', dtype=torch.bfloat16)
    mf = torch.randn(4096, 4096, device='cuda:0')
    torch.cuda.synchronize()
    for kind in AGGR.split(','):
        res = []
        for it in range(LITERS):
            with torch.cuda.stream(streams[1]):
                if kind == 'copy':
                    big2 = big.clone()
                elif kind == 'mfma':
                    mc = ma @ ma
                elif kind == 'valu':
                    mg = torch.sin(mf) * torch.cos(mf)
                elif kind == 'both':
                    big2 = big.clone(); mc = ma @ ma
            with torch.cuda.stream(streams[0]):
                res.append(layer_call(0, 'conv6'))
            if it % 16 == 15:
                streams[1].synchronize()
        torch.cuda.synchronize()
        print('  conv6 on engine 0 beside torch aggressor %-5s: mismatching runs %d of %d' % (kind, len(count_bad(res, refs['conv6'])), LITERS), flush=True)
        del res
SKIPS = os.environ.get('DET_SKIPS')      # comma-separated debug_skip masks: engine 1 repeats model() with those launch groups left out, engine 0 the victim layer
if SKIPS:
    vict = os.environ.get('DET_SKIP_VICTIM', 'conv5')
    VIT = int(os.environ.get('DET_VICTIM_ITERS', '300'))
    for mask in SKIPS.split(','):
        engs[1].set_option('debug_skip', int(mask))
        res = run_both(lambda e: layer_call(e, vict if e == 0 else 'model'), VIT)
        print('  victim %s beside model(debug_skip=%s): mismatching runs %d of %d' % (vict, mask, len(count_bad(res[0], refs[vict])), VIT), flush=True)
        del res
    engs[1].set_option('debug_skip', 0)
VICTIM = os.environ.get('DET_VICTIM')      # 'victim:aggressor' scopes: engine 0 repeats the first while engine 1 repeats the second; only engine 0 is checked
if VICTIM:
    va, vb = VICTIM.split(':')
    VIT = int(os.environ.get('DET_VICTIM_ITERS', '300'))
    res = run_both(lambda e: layer_call(e, va if e == 0 else vb), VIT)
    print('  victim %s beside %s: mismatching runs %d of %d' % (va, vb, len(count_bad(res[0], refs[va])), VIT), flush=True)
    del res
if PAIRS:
    names = list(refs)
    for a in names:
        row = []
        for b in names:
            res = run_both(lambda e: layer_call(e, a if e == 0 else b), LITERS)
            row.append('%d/%d' % (len(count_bad(res[0], refs[a])), len(count_bad(res[1], refs[b]))))
            del res
        print('  pair %-18s x %s: %s' % (a, [n.replace('conv', 'c').replace('_fullres', 'f').replace('_halfres', 'h').replace('_quarterres', 'q') for n in names], row), flush=True)

# ---- where do the differences sit?  (first differing run per engine, the layer with the most mismatches is the interesting one)
for scope in ('conv4_fullres', 'conv5', 'conv6'):
    if scope not in refs:
        continue
    res = run_both(lambda e: layer_call(e, scope), LITERS)
    ref = refs[scope].cpu().numpy()
    for e in range(2):
        nshown = 0
        for i, o in enumerate(res[e]):
            if not torch.equal(o, refs[scope]):
                d = o.cpu().numpy() != ref
                idx = np.argwhere(d)
                print(scope, 'engine', e, 'run', i, 'n', len(idx), 'images', np.unique(idx[:, 0]), 'rows', np.unique(idx[:, 1])[:12], 'cols', np.unique(idx[:, 2])[:12],
                      'channels', np.unique(idx[:, 3])[:16], '... n_ch', len(np.unique(idx[:, 3])), 'max|d|', float(np.abs(o.cpu().numpy() - ref).max()),
                      'max|ref|', float(np.abs(ref).max()))
                print('   first differing entries (image, row, col, channel: got, want):', [(tuple(int(v) for v in ix), float(o.cpu().numpy()[tuple(ix)]), float(ref[tuple(ix)])) for ix in idx[:6]])
                dd = np.abs(o.cpu().numpy().astype(np.float64) - ref)
                big = np.argwhere(dd > 1e-3)
                print('   |d| quantiles 50/90/99/99.9/max: %s; entries > 1e-3: %d, rows %s cols %s channels %s' % (
                    ['%.1e' % q for q in np.quantile(dd[dd > 0], [0.5, 0.9, 0.99, 0.999, 1.0])], len(big), np.unique(big[:, 1])[:40], np.unique(big[:, 2])[:40], np.unique(big[:, 3])))
                rowmax = dd.max(axis=(0, 2, 3))
                print('   max |d| per output row:', ['%.0e' % v for v in rowmax])
                band = np.where(rowmax > 1e-3)[0]
                if len(band):
                    dfull = o.cpu().numpy().astype(np.float64) - ref
                    bi = int(np.unique(big[:, 0])[0])
                    sp = np.abs(np.fft.rfft(dfull[bi][band], n=96, axis=1)).mean(axis=(0, 2))      # mean |rfft along x| over the band's rows and the channels
                    top = np.argsort(-sp)[:6]
                    print('   error spectrum along x (96-point bins) in the band: top bins', [(int(k), '%.1e' % sp[k]) for k in top], 'median bin %.1e' % np.median(sp))
                nshown += 1
                if nshown >= 3:
                    break
    del res
for en in engs:
    en.close()
print('done')
