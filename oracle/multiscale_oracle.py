"""NumPy restatement of the reference's multi-scale test-time evaluation:
`get_different_scales` (main.py:326-348), `scale_hm_back` (:351-379) and the per-image part of
`get_predictions` (:382-425).

TEST INFRASTRUCTURE (see oracle/__init__.py) -- PARITY UNPINNED: the reference calls
`skimage.transform.resize`, and scikit-image is neither vendored, pinned nor installed here.
The reference dates from Feb 2018 (main.py:444), i.e. scikit-image 0.13.x, whose `resize`
defaults are restated below: order=1 (bilinear), mode='constant', cval=0, clip=True,
preserve_range=False, no anti-aliasing filter; pixel centres at half-integers
(src = scale*(dst+0.5)-0.5); neighbours outside the image read as cval; the result is clipped
to the input's [min, max] (exact-cval pixels are kept when cval lies outside that range).
`resize_skimage` is the explicit formulation, `resize_skimage_scipy` an independent one on
`scipy.ndimage.map_coordinates(order=1, mode='grid-constant')`.
"""
import numpy as np
from scipy import ndimage

PAD_ARRAY = [1.1, 1.2, 1.3, 1.4]     # main.py:402
CROP_ARRAY = [0.7, 0.8, 0.9, 1.0]


def _coords(n_in, n_out):
    return (n_in / n_out) * (np.arange(n_out) + 0.5) - 0.5


def _clip_like_skimage(out, image, cval=0.0):
    mn, mx = image.min(), image.max()
    preserve = not (mn <= cval <= mx)
    mask = (out == cval) if preserve else None
    out = np.clip(out, mn, mx)
    if preserve:
        out[mask] = cval
    return out


def resize_skimage(image, out_h, out_w):
    """skimage.transform.resize(image, (out_h, out_w)) for [H,W,C] float input, 0.13.x defaults."""
    image = np.asarray(image, np.float64)
    H, W, _ = image.shape
    r, c = _coords(H, out_h), _coords(W, out_w)
    r0, r1 = np.floor(r).astype(int), np.ceil(r).astype(int)
    c0, c1 = np.floor(c).astype(int), np.ceil(c).astype(int)
    dr, dc = (r - r0)[:, None, None], (c - c0)[None, :, None]

    def px(ri, ci):
        ok = ((ri >= 0) & (ri <= H - 1))[:, None] & ((ci >= 0) & (ci <= W - 1))[None, :]
        v = image[np.clip(ri, 0, H - 1)][:, np.clip(ci, 0, W - 1)]
        return v * ok[:, :, None]

    top = (1 - dc) * px(r0, c0) + dc * px(r0, c1)
    bot = (1 - dc) * px(r1, c0) + dc * px(r1, c1)
    return _clip_like_skimage((1 - dr) * top + dr * bot, image)


def resize_skimage_scipy(image, out_h, out_w):
    image = np.asarray(image, np.float64)
    H, W, C = image.shape
    rr, cc = np.meshgrid(_coords(H, out_h), _coords(W, out_w), indexing='ij')
    out = np.stack([ndimage.map_coordinates(image[:, :, k], [rr, cc], order=1, mode='grid-constant', cval=0.0)
                    for k in range(C)], axis=2)
    return _clip_like_skimage(out, image)


def scale_windows(pad_array, crop_array, orig_h, orig_w):
    """(y0, x0, h, w) source windows of the 8 rescaled copies, main.py:328-341: a padded copy is
    a window that extends beyond the image (zeros outside), a cropped copy a window inside it.
    Python `round` (banker's) as in the reference."""
    wins = []
    for pad_c in pad_array:
        ph, pw = round(orig_h * (pad_c - 1) / 2), round(orig_w * (pad_c - 1) / 2)
        wins.append((-ph, -pw, orig_h + 2 * ph, orig_w + 2 * pw))
    for crop_c in crop_array:
        h1 = round((1 - crop_c) / 2 * orig_h)
        w1 = round((1 - crop_c) / 2 * orig_w)
        wins.append((h1, w1, round(crop_c * orig_h), round(crop_c * orig_w)))
    return wins


def back_windows(pad_array, crop_array, orig_h, orig_w):
    """Windows of `scale_hm_back` (main.py:353-369): the inverse crop / pad on the heat maps."""
    wins = []
    for pad_c in pad_array:
        crop_c = 1 / pad_c
        h1 = round((1 - crop_c) / 2 * orig_h)
        w1 = round((1 - crop_c) / 2 * orig_w)
        wins.append((h1, w1, round(crop_c * orig_h), round(crop_c * orig_w)))
    for crop_c in crop_array:
        pad_c = 1 / crop_c
        ph, pw = round(orig_h * (pad_c - 1) / 2), round(orig_w * (pad_c - 1) / 2)
        wins.append((-ph, -pw, orig_h + 2 * ph, orig_w + 2 * pw))
    return wins


def take_window(x, win):
    """np.lib.pad(..., 'constant', 0) / slicing of main.py:331,339 as one windowed read."""
    y0, x0, h, w = win
    H, W, C = x.shape
    out = np.zeros((h, w, C), x.dtype)
    ys, xs = max(y0, 0), max(x0, 0)
    ye, xe = min(y0 + h, H), min(x0 + w, W)
    out[ys - y0:ye - y0, xs - x0:xe - x0] = x[ys:ye, xs:xe]
    return out


def get_different_scales(x, pad_array=PAD_ARRAY, crop_array=CROP_ARRAY, orig_h=480, orig_w=720):
    """main.py:326-348 -> [8, orig_h, orig_w, 3] float64."""
    return np.array([resize_skimage(take_window(x, w), orig_h, orig_w) for w in scale_windows(pad_array, crop_array, orig_h, orig_w)])


def scale_hm_back(hms, pad_array=PAD_ARRAY, crop_array=CROP_ARRAY, orig_h=60, orig_w=90):
    """main.py:351-379 -> [8, orig_h, orig_w, K] float64."""
    wins = back_windows(pad_array, crop_array, orig_h, orig_w)
    return np.array([resize_skimage(take_window(hms[i], w), orig_h, orig_w) for i, w in enumerate(wins)])


def argmax_hm(hm):
    """main.py:389-397: [1,60,90,K] -> int [2,K] (row, col)."""
    hm = np.squeeze(hm)
    H, W, K = hm.shape
    raw = np.argmax(hm.reshape(H * W, K), axis=0)
    row = raw // W
    return np.stack([row, raw - row * W], axis=0)


def predict_one(x, y, forward):
    """One iteration of main.py:403-417.  `forward(x8, y8) -> (hm_pd, hm_sm)` stands for the
    sess.run at :406 (note the target maps `y` are repeated, not rescaled, :405)."""
    xs = get_different_scales(x)
    ys = np.repeat(np.expand_dims(y, 0), xs.shape[0], axis=0)
    hm_pd, hm_sm = forward(xs.astype(np.float32), ys)
    hm_pd = np.expand_dims(np.average(scale_hm_back(hm_pd), axis=0), 0)
    hm_sm = np.expand_dims(np.average(scale_hm_back(hm_sm), axis=0), 0)
    return argmax_hm(hm_pd), argmax_hm(hm_sm), hm_pd, hm_sm
