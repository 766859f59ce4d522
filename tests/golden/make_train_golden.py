"""Generate tests/golden/train_full.json: the full-size joint training step of one image restated by
oracle/train_oracle.py (float64 autograd; TensorFlow is unavailable -- parity unpinned), reduced to what a
GPU test can check in seconds: the four losses and, per trainable tensor, its max |g|, its L2 norm, 24
sampled entries, and `slack` = the largest difference between the float32 and the float64 run of the same
restatement (the arithmetic class of the reference is float32; a ReLU / max-pool decision that flips
between the two precisions moves a gradient by more than rounding).

    python tests/golden/make_train_golden.py        (about 10 minutes on 8 cores)
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import joint_cnn_mrf_amd  # noqa: E402,F401
from joint_cnn_mrf_amd import synth  # noqa: E402
from oracle import train_oracle as T  # noqa: E402

SEED_X, SEED_Y, B, LMBD = 77, 78, 1, 0.001
B_TOWER = 16      # BASELINE configs[4]: batch 128 over 8 GPUs = 16 images per tower (main.py:538-541,557-560)


def case(batch=B):
    p = synth.make_pd_params(debug=False, bn='trained')
    p.update(synth.make_sm_params(synth.synthetic_priors(), kind='trained'))
    return p, synth.make_images(batch, seed=SEED_X), synth.make_targets(batch, seed=SEED_Y)


def main(batch=B):
    """batch = 1 -> train_full.json / train_full_samples.npz; batch = 16 (python make_train_golden.py 16; about two hours on 8 cores:
    training-mode BatchNorm couples the images of a tower, so the 16-image golden cannot be assembled from smaller runs) ->
    train_full_b16.json / train_full_b16_samples.npz."""
    tag = '' if batch == B else '_b%d' % batch
    p, x, y = case(batch)
    r64 = T.loss_and_grads(x, y, p, use_sm=True, lmbd=LMBD)
    r32 = T.loss_and_grads(x, y, p, use_sm=True, lmbd=LMBD, dtype=torch.float32)
    rs = np.random.RandomState(5)
    out = {'losses': [r64['loss'], r64['loss_pd'], r64['loss_sm'], r64['l2']], 'tensors': {}}
    for k in sorted(r64['grads']):
        g = np.asarray(r64['grads'][k], np.float64).reshape(-1)
        g32 = np.asarray(r32['grads'][k], np.float64).reshape(-1)
        idx = np.unique(np.concatenate([rs.randint(0, g.size, 20), np.argsort(-np.abs(g))[:4]]))
        out['tensors'][k] = {'max': float(np.abs(g).max()), 'norm': float(np.linalg.norm(g)), 'slack': float(np.abs(g32 - g).max()),
                             'idx': [int(i) for i in idx], 'val': [float(g[i]) for i in idx]}
    # a denser sample for the mixed-precision test (cosine similarity per tensor): up to 512 seeded entries each
    rs2 = np.random.RandomState(6)
    dense = {}
    for k in sorted(r64['grads']):
        g = np.asarray(r64['grads'][k], np.float64).reshape(-1)
        idx = np.arange(g.size) if g.size <= 512 else np.sort(rs2.choice(g.size, 512, replace=False))
        dense[k + '|idx'] = idx.astype(np.int64)
        dense[k + '|val'] = g[idx].astype(np.float32)
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'train_full%s_samples.npz' % tag), **dense)
    mv = T.update_moving(p, r64['bn_stats'])
    out['moving'] = {k: {'idx': [0, int(v.size) - 1], 'val': [float(v.reshape(-1)[0]), float(v.reshape(-1)[-1])]} for k, v in mv.items()}
    with open(os.path.join(ROOT, 'tests', 'golden', 'train_full%s.json' % tag), 'w') as fh:
        json.dump(out, fh)
    print('wrote train_full%s.json: loss' % tag, out['losses'])


if __name__ == '__main__':
    main(int(sys.argv[1]) if len(sys.argv) > 1 else B)
