"""Pairwise prior builder: the data format on the input side of the spatial model.

Follows prepare_pairwise_distribution.py:13-14,29-48: for every ordered pair (joint,
cond_joint) a 120x180 histogram of heat-map-cell displacements over the training set,
centred at (60,90), normalised to sum 1 and smoothed with the 9x9 binomial kernel
(zero fill, 'same').  The reference reads the joint cells back out of `y_train_flic.npy`
as the arg-max cell of each 60x90 map (its lines 39-42); here the cells are the input.
"""
import numpy as np
from scipy import signal

from .synth import JOINT_NAMES

HM_HEIGHT, HM_WIDTH = 60, 90                                   # data.py:180
_COEFS = np.array([[1, 8, 28, 56, 70, 56, 28, 8, 1]], dtype=np.uint16) / 256   # :13
SMOOTH_KERNEL = _COEFS.T @ _COEFS                              # :14


def pairwise_distribution(cells_j, cells_c):
    """One smoothed displacement histogram (prepare_pairwise_distribution.py:29-48).
    cells_*: integer [N,2] (row, col) heat-map cells of the joint / conditioning joint."""
    pd = np.zeros([HM_HEIGHT * 2, HM_WIDTH * 2])
    dy = cells_j[:, 0].astype(np.int64) - cells_c[:, 0].astype(np.int64)
    dx = cells_j[:, 1].astype(np.int64) - cells_c[:, 1].astype(np.int64)
    np.add.at(pd, (HM_HEIGHT + dy, HM_WIDTH + dx), 1)          # :43 (one count per image)
    pd = pd / np.float32(np.sum(pd))                           # :44
    return signal.convolve2d(pd, SMOOTH_KERNEL, mode='same', boundary='fill', fillvalue=0)  # :45


def build_pairwise_distributions(cells, joint_names=JOINT_NAMES):
    """cells: [N,10,2] -> dict '<j>_<c>' -> float64 [120,180] for all 90 ordered pairs
    (prepare_pairwise_distribution.py:51-56), the content of pairwise_distribution.pickle."""
    out = {}
    for ji, j in enumerate(joint_names):
        for ci, c in enumerate(joint_names):
            if ci != ji:
                out[j + '_' + c] = pairwise_distribution(cells[:, ji], cells[:, ci])
    return out
