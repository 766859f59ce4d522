#!/usr/bin/env python
"""Generates the committed golden vectors.  Runs ONLY in the build container (it reads
/root/reference); the tests read the small .npy/.json files it writes next to itself.

  python tests/golden/make_golden.py [--exec-reference]

1. flic_train_cells.npy  [3987,10,2] uint8 -- heat-map cell (row, col) of the 9 joints + torso
   for every FLIC training example: the arg-max cell of the y_train heat maps, read back the way
   prepare_pairwise_distribution.py:39-42 reads it.  By default the maps come from this package's
   restatement (joint_cnn_mrf_amd.data) of data_FLIC.mat; with --exec-reference the reference's own
   data.py and prepare_pairwise_distribution.py are EXECUTED as well -- each in a child process inside a
   temp dir (they are untrusted upstream code; the JPEG reads of data.py are stood in for, its heat-map half only
   needs data_FLIC.mat) -- and their y_train / y_test arrays and the 90-pair pickle must equal the
   restatement's outputs bit for bit.
2. full_*.npy -- float64-oracle outputs of the FULL-SIZE network on 2 seeded images
   (inputs/weights are regenerated from seeds by the tests, only outputs are stored):
   pd_logits, sm_logits for FLIC priors with init / trained-like SM parameters, coords.
   The generator asserts a top-2 logit margin far above fp32 noise so argmax is well posed.
3. conv_mrf_*.npy -- one pair's pre-resize (61x91) and post-resize (60x90) maps.
4. layer_stats.json -- per-layer mean / abs-max of the full-size activations (for bisecting).
5. (--batch: writes ONLY these) batch64.npz / batch256.npz -- float64-oracle logits + coords of the images golden_util.BATCH64_GOLDEN /
   BATCH256_GOLDEN of the configs[1] / configs[2] test batches (golden_util.config_batch), so that those configurations are compared by VALUE at
   positions spread over the batch and not only on the golden pair; the top-2 margins are stored beside them (a dim image's margin is small:
   the tests compare coordinates where the margin is clear of fp32 noise and say how many joints that is).
"""
import json
import os
import pickle
import sys
import tempfile

import numpy as np
from scipy.io import loadmat

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = '/root/reference'

import joint_cnn_mrf_amd  # noqa: E402,F401
from joint_cnn_mrf_amd import priors, synth  # noqa: E402
from oracle import jcm_oracle as O  # noqa: E402

GOLDEN_SEEDS = dict(weights=7, images=2024, torso=2025, sm=11, conv6_gain=24.0)


_CHILD = r"""
import os, runpy, sys, types
import numpy as np
ref, script = sys.argv[1], sys.argv[2]
if script == 'data.py':
    # the script also reads 5003 JPEG frames that are not in the repository and imports plotting / image libraries that are not installed; only
    # those are stood in for (every frame reads as one black pixel -- the heat-map half never looks at the pixels)
    imageio = types.ModuleType('imageio'); imageio.imread = lambda path: np.zeros((1, 1, 3), np.uint8)
    skimage = types.ModuleType('skimage'); skimage.transform = types.ModuleType('skimage.transform')
    mpl = types.ModuleType('matplotlib'); mpl.pyplot = types.ModuleType('matplotlib.pyplot')
    sys.modules.update({'imageio': imageio, 'skimage': skimage, 'skimage.transform': skimage.transform, 'matplotlib': mpl, 'matplotlib.pyplot': mpl.pyplot})
    if not hasattr(np.lib, 'pad'):
        np.lib.pad = np.pad          # the alias the 2018 script uses; NumPy 2 dropped it
runpy.run_path(os.path.join(ref, script), run_name='__main__')
"""


def _run_reference_script(script, td):
    """Execute one of the reference's own scripts (untrusted upstream code) in a CHILD process whose working directory is the temp dir `td`:
    the stand-in modules, the NumPy alias and whatever the script writes stay in that process and that directory."""
    import subprocess
    subprocess.run([sys.executable, '-c', _CHILD, REF, script], cwd=td, check=True, stdout=subprocess.DEVNULL,
                   env={'PATH': os.environ.get('PATH', ''), 'HOME': td, 'PYTHONDONTWRITEBYTECODE': '1'})


def run_reference_data_script():
    """--exec-reference: /root/reference/data.py's own __main__ block on data_FLIC.mat -> (y_train, y_test) as IT builds them (the x_*_flic.npy arrays
    it writes are discarded).  Nothing of the script's text is kept here."""
    with tempfile.TemporaryDirectory() as td:
        os.symlink(os.path.join(REF, 'data_FLIC.mat'), os.path.join(td, 'data_FLIC.mat'))
        _run_reference_script('data.py', td)
        return np.load(os.path.join(td, 'y_train_flic.npy')), np.load(os.path.join(td, 'y_test_flic.npy'))


def cells_from_heat_maps(y):
    """prepare_pairwise_distribution.py:39-42: np.where(img == np.max(img)); must be unique."""
    n = y.shape[0]
    cells = np.zeros((n, 10, 2), np.uint8)
    flat = y.reshape(n, 60 * 90, 10)
    mx = flat.max(axis=1, keepdims=True)
    assert ((flat == mx).sum(axis=1) == 1).all(), 'a heat map has a non-unique maximum'
    idx = flat.argmax(axis=1)
    cells[:, :, 0] = idx // 90
    cells[:, :, 1] = idx % 90
    return cells


def run_reference_prior_builder(y_train):
    """--exec-reference: /root/reference/prepare_pairwise_distribution.py itself on y_train (child process, temp dir)."""
    with tempfile.TemporaryDirectory() as td:
        np.save(os.path.join(td, 'y_train_flic.npy'), y_train)
        _run_reference_script('prepare_pairwise_distribution.py', td)
        with open(os.path.join(td, 'pairwise_distribution.pickle'), 'rb') as fh:
            return pickle.load(fh)


def top2_margin(logits):
    flat = np.sort(logits.reshape(logits.shape[0], -1, logits.shape[3]), axis=1)
    return float((flat[:, -1] - flat[:, -2]).min())


def main():
    # Default: the heat maps and priors are built by this package's own restatement (joint_cnn_mrf_amd.data / .priors) from data_FLIC.mat.
    # --exec-reference additionally EXECUTES the reference's data.py and prepare_pairwise_distribution.py (child processes, temp dirs) and asserts
    # that their outputs equal the restatement's bit for bit -- the check that pins rows a10 / next-4 of SURVEY 8; the committed fixtures were
    # generated with it (round 3) and re-checked with it in round 4.
    exec_ref = '--exec-reference' in sys.argv
    from joint_cnn_mrf_amd import data as jdata
    xy, _names, is_train = jdata.load_flic(os.path.join(REF, 'data_FLIC.mat'))
    y_train = jdata.target_heat_maps(jdata.joint_cells(xy[is_train]))
    y_test = jdata.target_heat_maps(jdata.joint_cells(xy[~is_train]))
    if exec_ref:
        ry_train, ry_test = run_reference_data_script()
        assert np.array_equal(ry_train, y_train) and np.array_equal(ry_test, y_test)
        print('joint_cnn_mrf_amd.data: y_train / y_test identical to the arrays the reference data.py writes')
    print('y_train', y_train.shape, float(y_train.sum(axis=(1, 2)).mean()), 'y_test', y_test.shape)
    cells = cells_from_heat_maps(y_train)
    np.save(os.path.join(HERE, 'flic_train_cells.npy'), cells)
    np.save(os.path.join(HERE, 'flic_train_xy.npy'), xy[is_train])      # the annotations themselves (x, y of the nine joints, float64), input of joint_cnn_mrf_amd.data
    pri = priors.build_pairwise_distributions(cells)
    if exec_ref:
        ref = run_reference_prior_builder(y_train)
        assert sorted(ref) == sorted(pri) and len(ref) == 90
        for k in ref:
            assert ref[k].dtype == np.float64 and np.array_equal(ref[k], pri[k]), k
        print('priors: identical to the reference script output for all 90 pairs')
    del y_train

    g = GOLDEN_SEEDS
    p = synth.make_pd_params(debug=False, seed=g['weights'], bn='trained', conv6_gain=g['conv6_gain'])
    x = synth.make_images(2, seed=g['images'])
    torso = synth.make_torso(2, seed=g['torso'])
    taps = {}
    pd_logits = O.model(x, p, taps=taps)
    pd_prob = O.spatial_softmax(pd_logits)
    out = {'full_pd_logits': pd_logits, 'full_pd_coords': O.argmax_coords(pd_prob)}
    stats = {k: {'mean': float(v.mean()), 'absmax': float(np.abs(v).max()), 'shape': list(v.shape)} for k, v in taps.items()}
    stats['pd_top2_margin'] = top2_margin(pd_logits)
    stats['pd_prob_max'] = float(pd_prob.max())
    assert stats['pd_top2_margin'] > 1e-2, stats['pd_top2_margin']
    hm10 = np.concatenate([pd_prob, torso.astype(np.float64)], axis=3)
    for kind in ('init', 'trained'):
        sp = synth.make_sm_params(pri, kind=kind, seed=g['sm'])
        sm_logits = O.spatial_model(hm10, sp)
        out['full_sm_logits_' + kind] = sm_logits
        out['full_sm_coords_' + kind] = O.argmax_coords(O.spatial_softmax(sm_logits))
        stats['sm_top2_margin_' + kind] = top2_margin(sm_logits)
        assert stats['sm_top2_margin_' + kind] > 2e-3, (kind, stats['sm_top2_margin_' + kind])
        if kind == 'trained':
            key = 'lwri_lelb'
            prior = O.softplus5(np.asarray(sp['energy_' + key], np.float64))
            lik = O.softplus5(O.bn_infer(hm10, sp, 'bn_sm')[:, :, :, 1:2])
            out['conv_mrf_prior'] = prior
            out['conv_mrf_lik'] = lik
            out['conv_mrf_pre'] = O.conv_mrf_pre(prior, lik)
            out['conv_mrf_post'] = O.conv_mrf(prior, lik)
    for k, v in out.items():
        v = np.asarray(v)
        np.save(os.path.join(HERE, k + '.npy'), v.astype(np.float32) if v.dtype == np.float64 else v)
    stats['seeds'] = g
    with open(os.path.join(HERE, 'layer_stats.json'), 'w') as fh:
        json.dump(stats, fh, indent=1, sort_keys=True)
    print(json.dumps({k: v for k, v in stats.items() if 'margin' in k or 'max' in k}))


def main_batch():
    sys.path.insert(0, os.path.dirname(HERE))
    import golden_util as G
    pri = G.flic_priors()
    _x2, _t2, p = G.full_inputs()
    sp = synth.make_sm_params(pri, kind='trained', seed=GOLDEN_SEEDS['sm'])
    for B, idx in ((64, G.BATCH64_GOLDEN), (256, G.BATCH256_GOLDEN)):
        x, torso = G.config_batch(B)
        idx = np.asarray(idx)
        xs, ts = x[idx], torso[idx]
        del x
        pd_logits = np.concatenate([O.model(xs[i:i + 1], p) for i in range(len(idx))], axis=0)
        pd_prob = O.spatial_softmax(pd_logits)
        sm_logits = O.spatial_model(np.concatenate([pd_prob, ts.astype(np.float64)], axis=3), sp)

        def margins(lg):
            flat = np.sort(lg.reshape(lg.shape[0], -1, lg.shape[3]), axis=1)
            return flat[:, -1] - flat[:, -2]
        out = dict(idx=idx.astype(np.int32), pd_logits=pd_logits.astype(np.float32), sm_logits=sm_logits.astype(np.float32),
                   pd_coords=O.argmax_coords(pd_prob), sm_coords=O.argmax_coords(O.spatial_softmax(sm_logits)),
                   pd_margin=margins(pd_logits), sm_margin=margins(sm_logits))
        np.savez_compressed(os.path.join(HERE, 'batch%d.npz' % B), **out)
        print(B, 'pd margins (min per image)', out['pd_margin'].min(axis=1).round(4).tolist(), 'sm', out['sm_margin'].min(axis=1).round(4).tolist(),
              'pd logit scale', float(np.abs(pd_logits).max()))


if __name__ == '__main__':
    if '--batch' in sys.argv:
        main_batch()
    else:
        main()
