"""NumPy restatement of the reference's joint-heat-map inference forward.

TEST INFRASTRUCTURE (see oracle/__init__.py) -- PARITY UNPINNED: TensorFlow is absent, so
every op below restates documented TF-1.x semantics; citations are to /root/reference.

Layout everywhere: NHWC, images 480 rows x 720 cols, heat maps 60 rows x 90 cols
(data.py:10-12).  Parameters live in a flat dict keyed by the reference's TF variable
names:
    '<scope>/weights' [k,k,Cin,Cout] (HWIO, main.py:158), '<scope>/biases' [Cout],
    '<scope>/BatchNorm/{gamma,beta,moving_mean,moving_variance}' [Cout],
    'bn_sm/BatchNorm/*' [10], 'energy_<j>_<c>' [1,120,180,1], 'bias_<j>_<c>' [1,60,90,1].

All arithmetic runs in `dtype` (float64 for goldens; float32 for the timed CPU baseline).
The bilinear-resize sampling positions are always computed in float32, exactly as TF's
kernel does (`scale = in/float(out)`, `in = i*scale`), so that the interpolation weights
are the ones TF would use bit for bit.
"""
import numpy as np

# main.py:18 -- order defines channel ids AND the summation order in spatial_model.
JOINT_NAMES = ['lsho', 'lelb', 'lwri', 'rsho', 'relb', 'rwri', 'lhip', 'rhip', 'nose', 'torso']
# main.py:24-26 -- fully connected: every joint conditioned on all the others, in name order.
JOINT_DEPENDENCE = {j: [c for c in JOINT_NAMES if c != j] for j in JOINT_NAMES}
N_JOINTS = 9          # main.py:458
BN_EPS = 1e-3         # tf.contrib.layers.batch_norm default epsilon (main.py:113,129)
SM_DELTA = 10 ** -6   # main.py:110
SOFTPLUS_ALPHA = 5    # main.py:107
N_FILTERS = (64, 128, 256, 512, 512)   # main.py:38


def n_filters(debug=False):
    """main.py:38-41: `--debug` divides every filter count by 4."""
    return tuple(f // 4 for f in N_FILTERS) if debug else N_FILTERS


# --------------------------------------------------------------------------- TF-1.x ops
def same_padding(in_size, k, stride):
    """TF 'SAME': out=ceil(in/s); total=max((out-1)*s+k-in,0); before=total//2."""
    out = -(-in_size // stride)
    total = max((out - 1) * stride + k - in_size, 0)
    return out, total // 2, total - total // 2


def conv2d_same(x, w, stride, dtype=np.float64):
    """tf.nn.conv2d(x, W, [1,s,s,1], 'SAME') (main.py:133-135): NHWC x HWIO cross-correlation."""
    x = np.asarray(x, dtype)
    w = np.asarray(w, dtype)
    B, H, W, Cin = x.shape
    k = w.shape[0]
    Cout = w.shape[3]
    Ho, pt, pb = same_padding(H, k, stride)
    Wo, pl, pr = same_padding(W, k, stride)
    xp = np.zeros((B, H + pt + pb, W + pl + pr, Cin), dtype)
    xp[:, pt:pt + H, pl:pl + W, :] = x
    out = np.zeros((B * Ho * Wo, Cout), dtype)
    for ky in range(k):
        for kx in range(k):
            patch = xp[:, ky:ky + stride * (Ho - 1) + 1:stride, kx:kx + stride * (Wo - 1) + 1:stride, :]
            out += patch.reshape(B * Ho * Wo, Cin) @ w[ky, kx]
    return out.reshape(B, Ho, Wo, Cout)


def max_pool_same(x, size=2, stride=2):
    """tf.nn.max_pool(..., 'SAME') (main.py:172-174): padded cells never win (-inf)."""
    B, H, W, C = x.shape
    Ho, pt, pb = same_padding(H, size, stride)
    Wo, pl, pr = same_padding(W, size, stride)
    xp = np.full((B, H + pt + pb, W + pl + pr, C), -np.inf, x.dtype)
    xp[:, pt:pt + H, pl:pl + W, :] = x
    out = np.full((B, Ho, Wo, C), -np.inf, x.dtype)
    for dy in range(size):
        for dx in range(size):
            out = np.maximum(out, xp[:, dy:dy + stride * (Ho - 1) + 1:stride, dx:dx + stride * (Wo - 1) + 1:stride, :])
    return out


def resize_weights_tf1(out_size, in_size):
    """Sampling table of TF-1.x ResizeBilinear, align_corners=False (no half-pixel centres).

    scale = in/float(out) in float32; src = i*scale in float32; lower=floor(src),
    upper=min(lower+1, in-1); lerp = src-lower (float32).  (`ceil` in older kernels gives
    the same result because lerp==0 whenever src is integral.)
    """
    scale = np.float32(in_size) / np.float32(out_size)
    src = (np.arange(out_size, dtype=np.float32) * scale).astype(np.float32)
    lower = np.floor(src).astype(np.int64)
    upper = np.minimum(lower + 1, in_size - 1)
    lerp = (src - lower.astype(np.float32)).astype(np.float32)
    return lower, upper, lerp


def resize_bilinear_tf1(x, out_h, out_w):
    """tf.image.resize_images(x, [out_h,out_w]) (main.py:51,58,60,67,89): BILINEAR,
    align_corners=False, TF-1.x legacy coordinates; identity when the size is unchanged.
    Lerp along x first (top/bottom rows), then along y -- the kernel's order."""
    B, H, W, C = x.shape
    if (H, W) == (out_h, out_w):
        return x
    ylo, yhi, yl = resize_weights_tf1(out_h, H)
    xlo, xhi, xl = resize_weights_tf1(out_w, W)
    yl = yl.astype(x.dtype)[None, :, None, None]
    xl = xl.astype(x.dtype)[None, None, :, None]
    top = x[:, ylo][:, :, xlo] + (x[:, ylo][:, :, xhi] - x[:, ylo][:, :, xlo]) * xl
    bot = x[:, yhi][:, :, xlo] + (x[:, yhi][:, :, xhi] - x[:, yhi][:, :, xlo]) * xl
    return top + (bot - top) * yl


def bn_infer(x, p, scope, dtype=np.float64):
    """tf.contrib.layers.batch_norm, is_training=False: gamma*(x-mean)*rsqrt(var+1e-3)+beta."""
    g = np.asarray(p[scope + '/BatchNorm/gamma'], dtype)
    b = np.asarray(p[scope + '/BatchNorm/beta'], dtype)
    m = np.asarray(p[scope + '/BatchNorm/moving_mean'], dtype)
    v = np.asarray(p[scope + '/BatchNorm/moving_variance'], dtype)
    return (x - m) * (g / np.sqrt(v + dtype(BN_EPS))) + b


def tf_softplus(x):
    """tf.nn.softplus: log(exp(x)+1) with TF's shortcuts beyond +-(log(eps_f32)+2) ~ 13.94."""
    thr = np.log(np.finfo(np.float32).eps) + 2.0
    ex = np.exp(np.minimum(x, 50.0))
    return np.where(x > -thr, x, np.where(x < thr, ex, np.log1p(ex)))


def softplus5(x):
    """main.py:106-108: 1/5 * softplus(5x)."""
    return tf_softplus(SOFTPLUS_ALPHA * x) / SOFTPLUS_ALPHA


def bf16_round(x):
    """Round to the nearest bfloat16 (ties to even), returned in x's dtype.  Used by the `emulate='bf16'` mode below: the
    MI355X bf16 path (BASELINE.json configs[2]) keeps weights and activations in bf16 and accumulates in fp32, which
    the reference (fp32 throughout) does not do -- so the emulation is a statement about OUR kernels' arithmetic, used to
    hold them to rounding-level error instead of the 1e-2 distance between bf16 and fp32 results."""
    a = np.asarray(x)
    b = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    b = ((b + 0x7FFF + ((b >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    return b.view(np.float32).astype(a.dtype if a.dtype.kind == 'f' else np.float32).reshape(a.shape)


# --------------------------------------------------------------------------- model graph
def conv_layer(x, p, size, stride, name, last_layer=False, dtype=np.float64, emulate=None):
    """main.py:156-169: conv SAME + bias -> ReLU -> BN (BN *after* ReLU); last layer linear.
    emulate='bf16': input and weights rounded to bf16 before the (wide) accumulation, the result of a non-last layer
    rounded to bf16 (it is stored as a bf16 activation); bias / ReLU / BatchNorm stay in `dtype`."""
    w = p[name + '/weights']
    assert w.shape[0] == size and w.shape[1] == size
    if emulate == 'bf16':
        x, w = bf16_round(np.asarray(x, dtype)), bf16_round(np.asarray(w, dtype))
    elif emulate is not None:
        raise ValueError('emulate must be None or "bf16"')
    pre = conv2d_same(x, w, stride, dtype) + np.asarray(p[name + '/biases'], dtype)
    if last_layer:
        return pre
    out = bn_infer(np.maximum(pre, 0), p, name, dtype)
    return bf16_round(out) if emulate == 'bf16' else out


def model(x, p, n_joints=N_JOINTS, dtype=np.float64, taps=None, emulate=None):
    """main.py:29-74: 3-resolution part detector -> logits [B,60,90,n_joints].
    `taps` (dict) optionally receives every intermediate activation by scope name.
    emulate='bf16': every conv layer as conv_layer(..., emulate='bf16'), the merged map rounded to bf16 (max-pool and
    sub-sampling are exact on bf16 values) -- the arithmetic of Engine(precision='bf16')."""
    x = np.asarray(x, dtype)
    H, W = x.shape[1], x.shape[2]

    def keep(name, v):
        if taps is not None:
            taps[name] = v
        return v

    def branch(xin, res):
        h = keep('conv1_' + res, conv_layer(xin, p, 5, 2, 'conv1_' + res, dtype=dtype, emulate=emulate))
        h = keep('pool1_' + res, max_pool_same(h))
        h = keep('conv2_' + res, conv_layer(h, p, 5, 1, 'conv2_' + res, dtype=dtype, emulate=emulate))
        h = keep('pool2_' + res, max_pool_same(h))
        h = keep('conv3_' + res, conv_layer(h, p, 5, 1, 'conv3_' + res, dtype=dtype, emulate=emulate))
        h = keep('conv4_' + res, conv_layer(h, p, 9, 1, 'conv4_' + res, dtype=dtype, emulate=emulate))
        return h

    x1 = branch(x, 'fullres')                                               # main.py:43-49
    x2 = branch(resize_bilinear_tf1(x, H // 2, W // 2), 'halfres')          # main.py:51-57
    x2 = resize_bilinear_tf1(x2, x1.shape[1], x1.shape[2])                  # main.py:58
    x3 = branch(resize_bilinear_tf1(x, H // 4, W // 4), 'quarterres')       # main.py:60-66
    x3 = resize_bilinear_tf1(x3, x1.shape[1], x1.shape[2])                  # main.py:67
    h = (x1 + x2 + x3) / dtype(3)                                           # main.py:69-70
    h = keep('merge', bf16_round(h) if emulate == 'bf16' else h)
    h = keep('conv5', conv_layer(h, p, 9, 1, 'conv5', dtype=dtype, emulate=emulate))     # main.py:71
    return keep('conv6', conv_layer(h, p, 9, 1, 'conv6', last_layer=True, dtype=dtype, emulate=emulate))  # :72


def spatial_softmax(hm):
    """main.py:212-217: softmax over the H*W pixels of each (b,k) map (max-subtracted)."""
    B, H, W, K = hm.shape
    z = hm.reshape(B, H * W, K)
    z = z - z.max(axis=1, keepdims=True)
    e = np.exp(z)
    return (e / e.sum(axis=1, keepdims=True)).reshape(B, H, W, K)


def conv_mrf_pre(A, Bm, dtype=np.float64):
    """main.py:83-87: VALID cross-correlation of the 120x180 prior with the *flipped*
    60x90 likelihood of every image  ==  true 2-D convolution, output 61x91:
        Cpre[b,y,x] = sum_{u,v} A[y+59-u, x+89-v] * B[b,u,v].
    A: [1,120,180,1], Bm: [B,60,90,1] -> [B,61,91,1]."""
    A2 = np.asarray(A, dtype)[0, :, :, 0]
    B3 = np.asarray(Bm, dtype)[:, :, :, 0]
    nb, hb, wb = B3.shape
    Ho, Wo = A2.shape[0] - hb + 1, A2.shape[1] - wb + 1
    flt = B3[:, ::-1, ::-1].reshape(nb, hb * wb)                       # tf.reverse, main.py:84
    win = np.lib.stride_tricks.sliding_window_view(A2, (hb, wb))       # [Ho,Wo,hb,wb]
    out = win.reshape(Ho * Wo, hb * wb) @ flt.T                        # [Ho*Wo, B]
    return out.T.reshape(nb, Ho, Wo, 1)


def conv_mrf(A, Bm, hm_height=60, hm_width=90, dtype=np.float64):
    """main.py:77-91: conv_mrf_pre then bilinear resize 61x91 -> 60x90 (the crop at :88 is
    commented out in the reference; the resize at :89 is what runs)."""
    return resize_bilinear_tf1(conv_mrf_pre(A, Bm, dtype), hm_height, hm_width)


def spatial_model(heat_map, p, n_joints=N_JOINTS, dtype=np.float64):
    """main.py:94-125.  heat_map [B,60,90,10] (9 part-detector probabilities + torso map).
    E_j = log(sp(h_j)+d) + sum_{c != j, name order} log(conv_mrf(sp(e_{j|c}), sp(h_c)) + sp(b_{j|c}) + d)."""
    hm = bn_infer(np.asarray(heat_map, dtype), p, 'bn_sm', dtype)           # main.py:112-113
    hh, ww = hm.shape[1], hm.shape[2]
    delta = dtype(SM_DELTA)
    out = []
    for jid, jname in enumerate(JOINT_NAMES[:n_joints]):                    # main.py:114
        energy = np.log(softplus5(hm[:, :, :, jid:jid + 1]) + delta)        # main.py:117
        for cname in JOINT_DEPENDENCE[jname]:                               # main.py:118
            cid = JOINT_NAMES.index(cname)
            prior = softplus5(np.asarray(p['energy_%s_%s' % (jname, cname)], dtype))   # :120
            lik = softplus5(hm[:, :, :, cid:cid + 1])                                   # :121
            bias = softplus5(np.asarray(p['bias_%s_%s' % (jname, cname)], dtype))       # :122
            energy = energy + np.log(conv_mrf(prior, lik, hh, ww, dtype) + bias + delta)  # :123
        out.append(energy)
    return np.stack(out, axis=3)[:, :, :, :, 0]                             # main.py:125


def argmax_coords(hm):
    """evaluation.py:15-24 / main.py:389-397: first-occurrence flat argmax over H*W per
    (b,k); row = idx // W, col = idx - row*W.  Returns int32 [B,2,K]."""
    B, H, W, K = hm.shape
    idx = np.argmax(hm.reshape(B, H * W, K), axis=1)
    row = idx // W
    col = idx - row * W
    return np.stack([row, col], axis=1).astype(np.int32)


def forward(x, torso, p, use_sm=True, dtype=np.float64):
    """The graph of main.py:522-531 for one tower.  x [B,480,720,3]; torso [B,60,90,1] =
    y_in[..., 9:] (main.py:528).  Returns a dict with every stage output."""
    r = {}
    r['pd_logits'] = model(x, p, dtype=dtype)
    r['pd_prob'] = spatial_softmax(r['pd_logits'])                          # main.py:523
    r['pd_coords'] = argmax_coords(r['pd_prob'])
    if use_sm:
        hm10 = np.concatenate([r['pd_prob'], np.asarray(torso, dtype)], axis=3)   # main.py:528
        r['sm_logits'] = spatial_model(hm10, p, dtype=dtype)                # main.py:530
        r['sm_prob'] = spatial_softmax(r['sm_logits'])                      # main.py:531
        r['sm_coords'] = argmax_coords(r['sm_prob'])
    return r


def det_rate(hm_pred, hm_target, normalized_radius=10, joints='all'):
    """evaluation.py:4-37: percentage of (image, joint) pairs whose arg-max is within
    `normalized_radius` % of the torso length (target channels 0 and 7) of the target's arg-max.
    A zero torso length divides by zero exactly as the reference does (inf / nan compare False)."""
    pred = argmax_coords(hm_pred).astype(np.float32)
    true = argmax_coords(hm_target).astype(np.float32)
    torso = np.linalg.norm(true[:, :, 0] - true[:, :, 7], axis=1, keepdims=True)
    with np.errstate(divide='ignore', invalid='ignore'):
        nd = np.linalg.norm(pred - true, axis=1) * 100 / torso
    if joints != 'all':
        nd = nd[:, list(joints)]
    return float(100 * np.mean((nd <= normalized_radius).astype(np.float32)))
