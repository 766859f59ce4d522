"""Data preparation on the input side of the path (reference: data.py:88-196): FLIC annotations
(`data_FLIC.mat`) + JPEG frames -> the four arrays main.py loads (main.py:288-295):

    x_{train,test}_flic.npy  [n,480,720,3] float32 in [0,1]      (data.py:128-130)
    y_{train,test}_flic.npy  [n,60,90,10]  float32 target heat maps, channel 9 = torso (data.py:163-189)

Host-side NumPy only (this is file preparation, not the hot path).  The heat-map half needs just the
.mat file; the image half needs the FLIC JPEGs and Pillow.  The ICLR-2014 rescaling branch
(`iclr_data_preparation`, off in the reference, data.py:92) is not implemented.
"""
import os

import numpy as np

from .synth import JOINT_NAMES

ORIG_H, ORIG_W = 480, 720                 # data.py:119
HM_H, HM_W, HM_STRIDE = 60, 90, 8         # data.py:175,180
# column of each annotated point in the 2 x 29 `coords` matrix of a FLIC example (data.py:96-100)
FLIC_COLUMN = {'lsho': 0, 'lelb': 1, 'lwri': 2, 'rsho': 3, 'relb': 4, 'rwri': 5, 'lhip': 6, 'rhip': 9, 'nose': 16, 'torso': 28}
BLOB = np.outer([1, 2, 1], [1, 2, 1]).astype(np.float32) / 16      # data.py:112-114


def flip_backward_poses(pts):
    """data.py:35-49: a person seen from behind (left hip left of the right hip in the image) gets the
    left/right wrist, elbow, hip and shoulder annotations exchanged.  pts: dict name -> (x, y).
    The reference swaps through NumPy *views*, so after its first assignment both names hold the
    right-hand point: the net effect is `left := right` with the right side unchanged."""
    if pts['lhip'][0] < pts['rhip'][0]:
        for l, r in (('lwri', 'rwri'), ('lelb', 'relb'), ('lhip', 'rhip'), ('lsho', 'rsho')):
            pts[l] = pts[r]
    return pts


def joint_cells(xy):
    """xy: [n,2,9] image coordinates (x, y) of the nine joints in JOINT_NAMES order -> integer
    blob centres [n,10,2] (row, col; row 60 / col 90 possible for points clamped to the image edge, whose
    blob is then cut by the border) incl. the torso point (mean of shoulders and hips, data.py:165-167),
    after the backward-pose flip and the clamp to the image (data.py:171)."""
    xy = np.asarray(xy, np.float64)
    n = xy.shape[0]
    cells = np.zeros((n, 10, 2), np.int64)
    for i in range(n):
        pts = {name: xy[i, :, k].copy() for k, name in enumerate(JOINT_NAMES[:9])}
        pts = flip_backward_poses(pts)
        pts['torso'] = (pts['lsho'] + pts['rhip'] + pts['rsho'] + pts['lhip']) / 4
        for k, name in enumerate(JOINT_NAMES):
            x, y = pts[name]
            row = max(min(y, ORIG_H), 0) / HM_STRIDE
            col = max(min(x, ORIG_W), 0) / HM_STRIDE
            cells[i, k] = (int(row), int(col))           # int(coords + pad - 1) - pad + 1 of data.py:178-180
    return cells


def target_heat_maps(cells):
    """cells [n,10,2] -> y [n,60,90,10] float32: the 3x3 binomial blob centred on each joint's cell,
    clipped at the map border (the reference pastes into a 5-cell padded map and crops, data.py:176-183)."""
    cells = np.asarray(cells)
    n = cells.shape[0]
    m = 2                                                 # margin: a centre may sit on row 60 / column 90 (clamp to 480 / 720)
    y = np.zeros((n, HM_H + 2 * m, HM_W + 2 * m, cells.shape[1]), np.float32)
    for i in range(n):
        for k in range(cells.shape[1]):
            r, c = int(cells[i, k, 0]), int(cells[i, k, 1])
            if not (0 <= r <= HM_H and 0 <= c <= HM_W):
                raise ValueError('cell (%d, %d) outside the clamped range' % (r, c))
            y[i, r - 1 + m:r + 2 + m, c - 1 + m:c + 2 + m, k] = BLOB
    return np.ascontiguousarray(y[:, m:HM_H + m, m:HM_W + m])


def load_flic(mat_path):
    """`data_FLIC.mat` -> (xy [n,2,9] float64, file names, is_train bool [n])  (data.py:93-103)."""
    from scipy.io import loadmat
    ex = loadmat(mat_path)['examples'][0]
    cols = [FLIC_COLUMN[j] for j in JOINT_NAMES[:9]]
    xy = np.stack([np.asarray(e[2], np.float64)[:, cols] for e in ex])
    names = [str(e[3][0]) for e in ex]
    is_train = np.array([int(e[7][0, 0]) == 1 for e in ex])
    return xy, names, is_train


def load_image(path):
    """data.py:126-130: RGB bytes / 255 as float32 [480,720,3]."""
    from PIL import Image
    with Image.open(path) as im:
        a = np.asarray(im.convert('RGB'), np.float32) / 255
    if a.shape != (ORIG_H, ORIG_W, 3):
        raise ValueError('%s is %s, expected 480x720 RGB' % (path, a.shape))
    return a


def prepare(mat_path, images_dir=None, out_dir='.'):
    """Write y_{train,test}_flic.npy (and x_*.npy when `images_dir` holds the JPEGs), data.py:120-196."""
    xy, names, is_train = load_flic(mat_path)
    out = {}
    for split, mask in (('train', is_train), ('test', ~is_train)):
        y = target_heat_maps(joint_cells(xy[mask]))
        np.save(os.path.join(out_dir, 'y_%s_flic.npy' % split), y)
        out['y_' + split] = y.shape
        if images_dir is not None:
            x = np.stack([load_image(os.path.join(images_dir, n)) for n, m in zip(names, mask) if m])
            np.save(os.path.join(out_dir, 'x_%s_flic.npy' % split), x)
            out['x_' + split] = x.shape
    return out
