"""Parameter files in the reference's variable naming.

The reference saves/restores with `tf.train.Saver` (main.py:604,612,666); no checkpoint ships
with it and TensorFlow is not available here, so the interchange format is a NumPy `.npz`
whose keys are exactly the TF variable names:

    conv{1..4}_{fullres,halfres,quarterres}/{weights,biases}, conv5/..., conv6/...,
    <scope>/BatchNorm/{gamma,beta,moving_mean,moving_variance}, bn_sm/BatchNorm/*,
    energy_<j>_<c> [1,120,180,1], bias_<j>_<c> [1,60,90,1]

A TF-1.x user exports one with
    r = tf.train.load_checkpoint(path); np.savez(out, **{k: r.get_tensor(k) for k in r.get_variable_to_shape_map()})
(optimizer slots such as '<var>/Adam' and 'n_iters' are ignored on load).
"""
import numpy as np

from . import synth


def expected_shapes(debug=False, use_sm=True, n_joints=synth.N_JOINTS):
    """name -> shape of every variable the inference path reads (main.py:44-72,112,477-487)."""
    out = {}
    for scope, k, _s, cin, cout, last in synth.conv_scopes(debug, n_joints):
        out[scope + '/weights'] = (k, k, cin, cout)
        out[scope + '/biases'] = (cout,)
        if not last:
            for n in ('gamma', 'beta', 'moving_mean', 'moving_variance'):
                out['%s/BatchNorm/%s' % (scope, n)] = (cout,)
    if use_sm:
        for n in ('gamma', 'beta', 'moving_mean', 'moving_variance'):
            out['bn_sm/BatchNorm/' + n] = (n_joints + 1,)
        for key in synth.pair_keys(n_joints):
            out['energy_' + key] = (1, 120, 180, 1)
            out['bias_' + key] = (1, 60, 90, 1)
    return out


def validate(params, debug=False, use_sm=True):
    """Raise ValueError naming every missing or mis-shaped variable."""
    problems = []
    for name, shape in expected_shapes(debug, use_sm).items():
        if name not in params:
            problems.append('missing %s %s' % (name, shape))
        elif tuple(np.shape(params[name])) != shape:
            problems.append('%s has shape %s, expected %s' % (name, tuple(np.shape(params[name])), shape))
    if problems:
        raise ValueError('parameter file does not match the model: ' + '; '.join(problems[:8]) +
                         (' ... (%d problems)' % len(problems) if len(problems) > 8 else ''))
    return params


def save_npz(path, params):
    np.savez(path, **{k: np.asarray(v, np.float32) for k, v in params.items()})


def load_npz(path, debug=False, use_sm=True):
    """Read a `.npz` keyed by TF variable names; keep only what the inference path reads."""
    want = expected_shapes(debug, use_sm)
    with np.load(path) as z:
        params = {k: np.asarray(z[k], np.float32) for k in z.files if k in want}
    return validate(params, debug, use_sm)
