"""Round 5: are TORCH's kernels corrupted beside this library's MFMA kernels?

Round 4 found that packed-fp32 VALU results (v_pk_add/mul/fma_f32) can come out wrong in lanes 48-63 while a wave of one of this library's MFMA
kernels (LDS fragment reads feeding v_mfma) is resident on the same CU -- which needs kernels of two streams on the GPU at once.
The library itself contains no packed-fp32 instruction any more; round 4 tried torch kernels as AGGRESSORS only.  Here torch is the VICTIM:

  stream A (victim):    a chain of torch ops that hipcc vectorises (elementwise mul/add/fma on float32, softmax, layer_norm, a float2-style
                        complex multiply), each iteration's result compared bit for bit with the first run's (computed alone on an idle GPU)
  stream B (aggressor): an Engine with call_order = 0 repeating conv6 (fp32 handle: the 64 x 32-tile channel GEMM, round 4's strongest aggressor),
                        conv5 (64 x 128 tile) or the whole tower

    python tools/coresidency_torch.py            (environment: CT_ITERS victim iterations per aggressor, default 300)

Prints mismatching iterations per (victim op, aggressor); 0 everywhere = torch's kernels are not affected."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import joint_cnn_mrf_amd  # noqa: E402,F401
from golden_util import full_inputs  # noqa: E402
from joint_cnn_mrf_amd.engine import Engine  # noqa: E402

ITERS = int(os.environ.get('CT_ITERS', '300'))
dev = 'cuda:0'
x, _torso, p = full_inputs()
sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
eng = Engine(device=0, stream=sb, call_order=False).load_params(p)
xd = torch.as_tensor(x, device=dev)
a5 = torch.rand((2, 60, 90, 512), device=dev)
torch.cuda.synchronize()

g = torch.Generator(device=dev)
g.manual_seed(3)
N = 1 << 22
u, v, w = (torch.randn(N, device=dev, generator=g) for _ in range(3))
m2 = torch.randn((4096, 1024), device=dev, generator=g)
cz = torch.view_as_complex(torch.randn((N // 2, 2), device=dev, generator=g))
cw = torch.view_as_complex(torch.randn((N // 2, 2), device=dev, generator=g))

VICTIMS = {
    'mul_add': lambda: u * v + w,
    'addcmul': lambda: torch.addcmul(w, u, v, value=0.5),
    'softmax': lambda: torch.softmax(m2, dim=1),
    'layer_norm': lambda: torch.nn.functional.layer_norm(m2, (1024,)),
    'complex_mul': lambda: torch.view_as_real(cz * cw),
    'sum_rows': lambda: m2.sum(dim=1),
}
AGGRESSORS = {
    'conv6 (64x32 GEMM tile)': lambda: eng.conv_layer(a5, 'conv6', 1, last_layer=True, n_out=9),
    'conv5 (64x128 GEMM tile)': lambda: eng.conv_layer(a5, 'conv5', 1, n_out=512),
    'tower': lambda: eng.model(xd),
}

refs = {}
for name, fn in VICTIMS.items():      # alone on an idle GPU
    with torch.cuda.stream(sa):
        refs[name] = fn().clone()
torch.cuda.synchronize()
for name, fn in VICTIMS.items():      # ... and reproducible alone
    with torch.cuda.stream(sa):
        assert torch.equal(fn(), refs[name]), 'victim %s is not reproducible on an idle GPU' % name
torch.cuda.synchronize()

total_bad = 0
for an, agg in AGGRESSORS.items():
    for vn, vic in VICTIMS.items():
        bad = torch.zeros((), dtype=torch.int64, device=dev)
        for i in range(ITERS):
            with torch.cuda.stream(sb):
                agg()
                if i % 4 == 0:
                    agg()
            with torch.cuda.stream(sa):
                r = vic()
                bad += (~torch.eq(r, refs[vn])).any().to(torch.int64)      # compared on the device, on the victim's stream
        torch.cuda.synchronize()
        nb = int(bad.item())
        total_bad += nb
        print('victim %-12s beside %-26s mismatching iterations %d of %d' % (vn, an, nb, ITERS), flush=True)
eng.close()
print('TOTAL mismatching iterations: %d' % total_bad)
