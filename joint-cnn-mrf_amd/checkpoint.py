"""Parameter files in the reference's variable naming.

The reference saves/restores with `tf.train.Saver` (main.py:604,612,666); no checkpoint ships
with it and TensorFlow is not available here, so the interchange format is a NumPy `.npz`
whose keys are exactly the TF variable names:

    conv{1..4}_{fullres,halfres,quarterres}/{weights,biases}, conv5/..., conv6/...,
    <scope>/BatchNorm/{gamma,beta,moving_mean,moving_variance}, bn_sm/BatchNorm/*,
    energy_<j>_<c> [1,120,180,1], bias_<j>_<c> [1,60,90,1]

A TF-1.x user exports one with
    r = tf.train.load_checkpoint(path); np.savez(out, **{k: r.get_tensor(k) for k in r.get_variable_to_shape_map()})
`load_npz` keeps the variables the inference path reads; `session_state` / `restore_session_state` add what
tf.train.Saver also saves and a resumed training run needs (main.py:604,612,666): the optimizer slots '<var>/Adam',
'<var>/Adam_1' (or '<var>/Momentum'), Adam's 'beta1_power' / 'beta2_power' and the update counter 'n_iters'.
The TensorFlow checkpoint-V2 files themselves (`<prefix>.index` + `<prefix>.data-00000-of-00001`) are read and written
by `tf_checkpoint.py`.
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


# ---------------------------------------------------------------------------------------------------- training sessions
ADAM_BETA1, ADAM_BETA2 = 0.9, 0.999          # tf.train.AdamOptimizer defaults (main.py:502)


def session_state(trainer, like):
    """Everything `saver.save(sess, ...)` writes for a training session (main.py:604,666), keyed by the TF variable names:
    the model variables (shapes taken from the dict `like`), the optimizer slots of every trainable variable, Adam's beta
    powers and n_iters."""
    import ctypes
    from . import _lib
    out = dict(trainer.get_params(like))
    n = ctypes.c_int64()
    slots = []
    for slot in (0, 1):
        buf = np.empty(trainer.n_elements, np.float32)
        _lib.check(trainer._lib.jcm_train_get_state(trainer.eng._h, slot, ctypes.c_void_p(buf.ctypes.data), buf.size, ctypes.byref(n)),
                   'jcm_train_get_state')
        slots.append(buf)
    adam = trainer.optimizer == 'adam'
    for name, off, cnt in trainer.layout:
        shape = np.asarray(like[name]).shape
        if adam:
            out[name + '/Adam'] = slots[0][off:off + cnt].reshape(shape).copy()
            out[name + '/Adam_1'] = slots[1][off:off + cnt].reshape(shape).copy()
        else:
            out[name + '/Momentum'] = slots[0][off:off + cnt].reshape(shape).copy()
    if adam:
        # tf.train.AdamOptimizer creates the accumulators AT beta (not 1) and multiplies them by beta in _finish after every
        # update: after n updates a Saver checkpoint holds beta ** (n + 1)
        out['beta1_power'] = np.float32(ADAM_BETA1 ** (n.value + 1))
        out['beta2_power'] = np.float32(ADAM_BETA2 ** (n.value + 1))
    out['n_iters'] = np.int32(n.value)
    return out


def restore_session_state(trainer, state):
    """`saver.restore` for the optimizer side: the model variables of `state` are loaded through Engine.load_params /
    jcm_update_tensor by the caller; this puts the slots and n_iters back (missing slots restore as zeros, like a
    checkpoint written before the optimizer existed)."""
    import ctypes
    from . import _lib
    adam = trainer.optimizer == 'adam'
    names = ('/Adam', '/Adam_1') if adam else ('/Momentum',)
    n_iters = int(state.get('n_iters', 0))
    for slot, suffix in enumerate(names):
        buf = np.zeros(trainer.n_elements, np.float32)
        for name, off, cnt in trainer.layout:
            if name + suffix in state:
                v = np.asarray(state[name + suffix], np.float32).reshape(-1)
                if v.size != cnt:
                    raise ValueError('%s has %d elements, expected %d' % (name + suffix, v.size, cnt))
                buf[off:off + cnt] = v
        _lib.check(trainer._lib.jcm_train_set_state(trainer.eng._h, slot, ctypes.c_void_p(buf.ctypes.data), buf.size, n_iters),
                   'jcm_train_set_state')
    if not adam:
        _lib.check(trainer._lib.jcm_train_set_state(trainer.eng._h, 1, None, 0, n_iters), 'jcm_train_set_state')
    return n_iters
