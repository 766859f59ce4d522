"""Second, independently written formulation of the oracle (torch-CPU ops + SciPy).

TEST INFRASTRUCTURE (see oracle/__init__.py) -- PARITY UNPINNED.  Where `jcm_oracle.py`
spells every op out with NumPy slicing and matmuls, this file leans on library ops with
*explicit* handling of the TF-1.x corner cases: `F.pad` for the asymmetric SAME padding,
`F.max_pool2d` on a -inf padded tensor, gather-based legacy bilinear, and
`scipy.signal.convolve2d(prior, likelihood, 'valid')` for conv_mrf (a true convolution,
which is what main.py:83-87 builds out of transpose + reverse + VALID correlation).
It doubles as the timed CPU baseline of bench.py (fp32, oneDNN), labelled
"CPU restatement (TF unavailable)".
"""
import math

import numpy as np
import torch
import torch.nn.functional as F
from scipy.signal import convolve2d

from .jcm_oracle import JOINT_NAMES, JOINT_DEPENDENCE, N_JOINTS, BN_EPS, SM_DELTA, SOFTPLUS_ALPHA


def _same_pad(n, k, s):
    out = math.ceil(n / s)
    total = max((out - 1) * s + k - n, 0)
    return total // 2, total - total // 2


def _t(a, dtype):
    return torch.as_tensor(np.ascontiguousarray(a)).to(dtype)


def conv2d_same(x, w_hwio, stride):
    """x: NCHW tensor; w: HWIO numpy/tensor.  main.py:133-135."""
    w = w_hwio.permute(3, 2, 0, 1).contiguous()
    k = w.shape[2]
    pt, pb = _same_pad(x.shape[2], k, stride)
    pl, pr = _same_pad(x.shape[3], k, stride)
    return F.conv2d(F.pad(x, (pl, pr, pt, pb)), w, stride=stride)


def max_pool_same(x):
    """main.py:172-174."""
    pt, pb = _same_pad(x.shape[2], 2, 2)
    pl, pr = _same_pad(x.shape[3], 2, 2)
    return F.max_pool2d(F.pad(x, (pl, pr, pt, pb), value=float('-inf')), 2, 2)


def resize_bilinear_tf1(x, oh, ow):
    """TF-1.x legacy bilinear (no half-pixel offset) on an NCHW tensor, via gathers."""
    H, W = x.shape[2], x.shape[3]
    if (H, W) == (oh, ow):
        return x

    def table(o, n):
        scale = torch.tensor(n, dtype=torch.float32) / torch.tensor(o, dtype=torch.float32)
        src = torch.arange(o, dtype=torch.float32) * scale
        lo = src.floor().long()
        hi = torch.clamp(lo + 1, max=n - 1)
        return lo, hi, (src - lo.float()).to(x.dtype)

    ylo, yhi, yl = table(oh, H)
    xlo, xhi, xl = table(ow, W)
    rows_lo, rows_hi = x[:, :, ylo], x[:, :, yhi]
    top = rows_lo[:, :, :, xlo] + (rows_lo[:, :, :, xhi] - rows_lo[:, :, :, xlo]) * xl
    bot = rows_hi[:, :, :, xlo] + (rows_hi[:, :, :, xhi] - rows_hi[:, :, :, xlo]) * xl
    return top + (bot - top) * yl[:, None]


def _bn(x, p, scope, dtype):
    g, b, m, v = (_t(p['%s/BatchNorm/%s' % (scope, n)], dtype) for n in
                  ('gamma', 'beta', 'moving_mean', 'moving_variance'))
    sh = (1, -1, 1, 1)
    return (x - m.view(sh)) * (g * torch.rsqrt(v + BN_EPS)).view(sh) + b.view(sh)


def conv_layer(x, p, stride, name, dtype, last_layer=False):
    """main.py:156-169."""
    z = conv2d_same(x, _t(p[name + '/weights'], dtype), stride) + _t(p[name + '/biases'], dtype).view(1, -1, 1, 1)
    return z if last_layer else _bn(F.relu(z), p, name, dtype)


def model(x_nhwc, p, dtype=torch.float64):
    """main.py:29-74 -> logits NHWC numpy."""
    x = _t(x_nhwc, dtype).permute(0, 3, 1, 2).contiguous()
    H, W = x.shape[2], x.shape[3]

    def branch(h, res):
        h = max_pool_same(conv_layer(h, p, 2, 'conv1_' + res, dtype))
        h = max_pool_same(conv_layer(h, p, 1, 'conv2_' + res, dtype))
        h = conv_layer(h, p, 1, 'conv3_' + res, dtype)
        return conv_layer(h, p, 1, 'conv4_' + res, dtype)

    x1 = branch(x, 'fullres')
    x2 = resize_bilinear_tf1(branch(resize_bilinear_tf1(x, H // 2, W // 2), 'halfres'), x1.shape[2], x1.shape[3])
    x3 = resize_bilinear_tf1(branch(resize_bilinear_tf1(x, H // 4, W // 4), 'quarterres'), x1.shape[2], x1.shape[3])
    h = (x1 + x2 + x3) / 3
    h = conv_layer(h, p, 1, 'conv5', dtype)
    h = conv_layer(h, p, 1, 'conv6', dtype, last_layer=True)
    return h.permute(0, 2, 3, 1).contiguous().numpy()


def spatial_softmax(hm):
    """main.py:212-217 on NHWC numpy."""
    t = torch.as_tensor(hm)
    B, H, W, K = t.shape
    return torch.softmax(t.reshape(B, H * W, K), dim=1).reshape(B, H, W, K).numpy()


def softplus5(t):
    return F.softplus(t * SOFTPLUS_ALPHA, beta=1.0, threshold=13.942385) / SOFTPLUS_ALPHA


def conv_mrf(A, Bm, hm_height=60, hm_width=90):
    """main.py:77-91 with SciPy's true 2-D convolution, one image at a time."""
    A2 = np.asarray(A)[0, :, :, 0]
    pre = np.stack([convolve2d(A2, np.asarray(Bm)[b, :, :, 0], mode='valid') for b in range(Bm.shape[0])])
    pre = torch.as_tensor(pre)[:, None]                         # [B,1,61,91]
    return resize_bilinear_tf1(pre, hm_height, hm_width).permute(0, 2, 3, 1).numpy()


def spatial_model(heat_map, p, n_joints=N_JOINTS, dtype=torch.float64):
    """main.py:94-125 on NHWC numpy input."""
    hm = _bn(_t(heat_map, dtype).permute(0, 3, 1, 2), p, 'bn_sm', dtype)      # NCHW
    out = []
    for jid, jname in enumerate(JOINT_NAMES[:n_joints]):
        e = torch.log(softplus5(hm[:, jid]) + SM_DELTA)
        for cname in JOINT_DEPENDENCE[jname]:
            cid = JOINT_NAMES.index(cname)
            prior = softplus5(_t(p['energy_%s_%s' % (jname, cname)], dtype)).numpy()
            lik = softplus5(hm[:, cid]).numpy()[..., None]
            bias = softplus5(_t(p['bias_%s_%s' % (jname, cname)], dtype))[0, :, :, 0]
            c = torch.as_tensor(conv_mrf(prior, lik))[:, :, :, 0].to(dtype)
            e = e + torch.log(c + bias + SM_DELTA)
        out.append(e)
    return torch.stack(out, dim=3).numpy()


def argmax_coords(hm):
    """evaluation.py:15-24."""
    t = torch.as_tensor(hm)
    B, H, W, K = t.shape
    idx = torch.argmax(t.reshape(B, H * W, K), dim=1)
    row = idx // W
    return torch.stack([row, idx - row * W], dim=1).to(torch.int32).numpy()


def forward(x, torso, p, use_sm=True, dtype=torch.float64):
    """main.py:522-531."""
    r = {'pd_logits': model(x, p, dtype)}
    r['pd_prob'] = spatial_softmax(r['pd_logits'])
    r['pd_coords'] = argmax_coords(r['pd_prob'])
    if use_sm:
        hm10 = np.concatenate([r['pd_prob'], np.asarray(torso, r['pd_prob'].dtype)], axis=3)
        r['sm_logits'] = spatial_model(hm10, p, dtype=dtype)
        r['sm_prob'] = spatial_softmax(r['sm_logits'])
        r['sm_coords'] = argmax_coords(r['sm_prob'])
    return r
