"""Multi-scale test-time evaluation: the caller of the inference tower in the reference
(`get_different_scales`, `scale_hm_back`, `get_predictions`, main.py:326-425).

The reference rescales on the host with scikit-image and feeds 8 copies of ONE image per
`sess.run`; here the 8 pad/crop + resize copies are produced on the device
(`jcm_window_resize`), any number of images share a forward (8 copies each), and the heat
maps are scaled back, averaged and arg-maxed on the device too.  Same names, arguments and
array layouts as the reference; the rounding of the window edges is Python's `round`, as there.
"""
import numpy as np
import torch

PAD_ARRAY = [1.1, 1.2, 1.3, 1.4]      # main.py:402
CROP_ARRAY = [0.7, 0.8, 0.9, 1.0]


def _scale_windows(pad_array, crop_array, orig_h, orig_w):
    wins = []
    for pad_c in pad_array:                                                   # main.py:328-333
        n_pad_h = round(orig_h * (pad_c - 1) / 2)
        n_pad_w = round(orig_w * (pad_c - 1) / 2)
        wins.append((-n_pad_h, -n_pad_w, orig_h + 2 * n_pad_h, orig_w + 2 * n_pad_w))
    for crop_c in crop_array:                                                 # main.py:334-341
        h1 = round((1 - crop_c) / 2 * orig_h)
        h2 = h1 + round(crop_c * orig_h)
        w1 = round((1 - crop_c) / 2 * orig_w)
        w2 = w1 + round(crop_c * orig_w)
        wins.append((h1, w1, h2 - h1, w2 - w1))
    return wins


def _back_windows(pad_array, crop_array, orig_h, orig_w):
    wins = []
    for crop_c in pad_array:                                                  # main.py:353-361
        crop_c = 1 / crop_c
        h1 = round((1 - crop_c) / 2 * orig_h)
        h2 = h1 + round(crop_c * orig_h)
        w1 = round((1 - crop_c) / 2 * orig_w)
        w2 = w1 + round(crop_c * orig_w)
        wins.append((h1, w1, h2 - h1, w2 - w1))
    for pad_c in crop_array:                                                  # main.py:363-369
        pad_c = 1 / pad_c
        n_pad_h = round(orig_h * (pad_c - 1) / 2)
        n_pad_w = round(orig_w * (pad_c - 1) / 2)
        wins.append((-n_pad_h, -n_pad_w, orig_h + 2 * n_pad_h, orig_w + 2 * n_pad_w))
    return wins


def get_different_scales(engine, x, pad_array, crop_array, orig_h, orig_w):
    """main.py:326-348.  x [H,W,3] or [N,H,W,3] device tensor -> [N*8, orig_h, orig_w, 3]
    (the 8 copies of an image are consecutive, pads first, then crops)."""
    if x.dim() == 3:
        x = x.unsqueeze(0)
    wins = _scale_windows(pad_array, crop_array, orig_h, orig_w)
    windows = [(i,) + w for i in range(x.shape[0]) for w in wins]
    return engine.window_resize(x.contiguous(), windows, orig_h, orig_w)


def scale_hm_back(engine, hms, pad_array, crop_array, orig_h, orig_w):
    """main.py:351-379.  hms [N*8, 60, 90, K] (copy s of image i at i*8+s) -> same shape."""
    wins = _back_windows(pad_array, crop_array, orig_h, orig_w)
    ns = len(wins)
    if hms.shape[0] % ns:
        raise ValueError('expected a multiple of %d heat maps' % ns)
    windows = [(i,) + wins[i % ns] for i in range(hms.shape[0])]
    return engine.window_resize(hms.contiguous(), windows, orig_h, orig_w)


def get_predictions(engine, X, Y, use_sm=True, images_per_forward=8):
    """main.py:382-425 without the det_rate bookkeeping: for every image the 8 rescaled copies go
    through the tower (with the UNSCALED target maps repeated, main.py:405), the heat maps are
    scaled back and averaged, and the arg-max coordinates are returned as the reference returns
    them: int arrays [2, K, N] (row, col) for the part detector and the spatial model."""
    dev = engine.device
    K = engine.n_joints
    n = X.shape[0]
    in_h, in_w = int(X.shape[1]), int(X.shape[2])
    pd_all, sm_all = [], []
    for i0 in range(0, n, images_per_forward):
        x = torch.as_tensor(np.ascontiguousarray(X[i0:i0 + images_per_forward], dtype=np.float32), device=dev)
        y = torch.as_tensor(np.ascontiguousarray(Y[i0:i0 + images_per_forward], dtype=np.float32), device=dev)
        m = x.shape[0]
        xs = get_different_scales(engine, x, PAD_ARRAY, CROP_ARRAY, in_h, in_w)                 # [m*8,480,720,3]
        torso = y[:, :, :, K:].repeat_interleave(8, dim=0).contiguous()                           # main.py:405,528
        r = engine.forward(xs, torso if use_sm else None, use_sm=use_sm, want_prob=True)
        hm_pd = engine.group_mean(scale_hm_back(engine, r['pd_prob'], PAD_ARRAY, CROP_ARRAY, 60, 90), 8)   # :407,413
        pd_all.append(engine.argmax_coords(hm_pd))                                                # :416
        if use_sm:
            hm_sm = engine.group_mean(scale_hm_back(engine, r['sm_prob'], PAD_ARRAY, CROP_ARRAY, 60, 90), 8)
            sm_all.append(engine.argmax_coords(hm_sm))
        assert m == hm_pd.shape[0]
    pd = torch.cat(pd_all).cpu().numpy().transpose(1, 2, 0)                                       # [2,K,N], :425
    sm = torch.cat(sm_all).cpu().numpy().transpose(1, 2, 0) if use_sm else pd
    return pd, sm
