"""Deterministic synthetic inputs and parameters for the joint-heat-map path.

There is no network for FLIC images or trained checkpoints, so tests and `bench.py` use
seeded data of the reference's shapes and initial distributions:

* conv weights: He truncated normal, sigma = sqrt(2/(k*k*Cin)), HWIO   (main.py:138-147)
* conv biases: 0                                                        (main.py:150-153)
* BatchNorm: contrib defaults gamma=1, beta=0, mean=0, var=1 ('identity'), or a
  'trained'-like set (gamma~U[.5,1.5], beta,mean~N(0,.1), var~U[.5,1.5])
* pairwise energies: float32 cast of the prior histograms               (main.py:482-484)
* pairwise biases: 1e-5                                                  (main.py:486-487)
* images: U[0,1) float32 [B,480,720,3]                                   (data.py:129-130)
* torso map: the 3x3 binomial blob [1,2,1]^T[1,2,1]/16                   (data.py:112-114,180-186)

Everything is keyed by the reference's TF variable names so a real checkpoint would map 1:1.
"""
import numpy as np
from scipy.special import ndtri

JOINT_NAMES = ['lsho', 'lelb', 'lwri', 'rsho', 'relb', 'rwri', 'lhip', 'rhip', 'nose', 'torso']  # main.py:18
N_JOINTS = 9                                                                                     # main.py:458
N_FILTERS = (64, 128, 256, 512, 512)                                                             # main.py:38
RESOLUTIONS = ('fullres', 'halfres', 'quarterres')

_PHI_M2 = 0.022750131948179195   # Phi(-2)
_PHI_P2 = 0.9772498680518208     # Phi(+2)


def n_filters(debug=False):
    """main.py:38-41."""
    return tuple(f // 4 for f in N_FILTERS) if debug else N_FILTERS


def conv_scopes(debug=False, n_joints=N_JOINTS):
    """[(scope, k, stride, Cin, Cout, last_layer)] in graph order (main.py:44-72)."""
    f = n_filters(debug)
    out = []
    for res in RESOLUTIONS:
        out += [('conv1_' + res, 5, 2, 3, f[0], False), ('conv2_' + res, 5, 1, f[0], f[1], False),
                ('conv3_' + res, 5, 1, f[1], f[2], False), ('conv4_' + res, 9, 1, f[2], f[3], False)]
    out += [('conv5', 9, 1, f[3], f[4], False), ('conv6', 9, 1, f[4], n_joints, True)]
    return out


def truncated_normal(rs, shape, stddev):
    """tf.truncated_normal (main.py:146): N(0, stddev) restricted to +-2 sigma.  Drawn by
    inverse CDF from one uniform stream so the sequence is a pure function of the seed."""
    u = rs.random_sample(int(np.prod(shape)))
    z = ndtri(_PHI_M2 + u * (_PHI_P2 - _PHI_M2))
    return (z * stddev).astype(np.float32).reshape(shape)


def make_pd_params(debug=False, seed=7, bn='identity', conv6_gain=1.0, n_joints=N_JOINTS):
    """Part-detector parameters under the reference's variable names."""
    rs = np.random.RandomState(seed)
    p = {}
    for scope, k, _s, cin, cout, last in conv_scopes(debug, n_joints):
        w = truncated_normal(rs, (k, k, cin, cout), np.sqrt(2.0 / (k * k * cin)))   # main.py:141,146
        if last and conv6_gain != 1.0:
            w = (w * np.float32(conv6_gain)).astype(np.float32)
        p[scope + '/weights'] = w
        p[scope + '/biases'] = np.zeros(cout, np.float32)                             # main.py:152
        if not last:
            _add_bn(p, scope, cout, rs, bn)
    return p


def _add_bn(p, scope, c, rs, kind):
    if kind == 'identity':
        g, b, m, v = np.ones(c), np.zeros(c), np.zeros(c), np.ones(c)
    elif kind == 'trained':
        g = rs.uniform(0.5, 1.5, c)
        b = rs.normal(0, 0.1, c)
        m = rs.normal(0, 0.1, c)
        v = rs.uniform(0.5, 1.5, c)
    else:
        raise ValueError(kind)
    for n, a in (('gamma', g), ('beta', b), ('moving_mean', m), ('moving_variance', v)):
        p['%s/BatchNorm/%s' % (scope, n)] = np.asarray(a, np.float32)


def pair_keys(n_joints=N_JOINTS):
    """The 81 '<j>_<c>' keys in graph order (main.py:479-481)."""
    return ['%s_%s' % (j, c) for j in JOINT_NAMES[:n_joints] for c in JOINT_NAMES if c != j]


def make_sm_params(priors, kind='init', seed=11, n_joints=N_JOINTS):
    """Spatial-model parameters.  `priors`: dict '<j>_<c>' -> [120,180] (pairwise_distribution
    pickle, main.py:297-299).  kind='init' reproduces main.py:477-487 exactly; 'trained' scales
    the energies/biases and gives bn_sm a non-trivial affine so the pairwise terms have real
    dynamic range (at init they are nearly constant)."""
    rs = np.random.RandomState(seed)
    p = {}
    for key in pair_keys(n_joints):
        e = np.asarray(priors[key], np.float32)                                       # main.py:482
        if kind == 'init':
            b = np.full((60, 90), 1e-5, np.float32)                                   # main.py:486
        else:
            e = (e * np.float32(400.0) - np.float32(0.05)).astype(np.float32)
            b = (0.02 * rs.random_sample((60, 90))).astype(np.float32)
        p['energy_' + key] = e.reshape(1, e.shape[0], e.shape[1], 1)                   # main.py:483
        p['bias_' + key] = b.reshape(1, 60, 90, 1)
    if kind == 'init':
        _add_bn(p, 'bn_sm', 10, rs, 'identity')
    else:
        p['bn_sm/BatchNorm/gamma'] = rs.uniform(20.0, 40.0, 10).astype(np.float32)
        p['bn_sm/BatchNorm/beta'] = rs.normal(-0.05, 0.02, 10).astype(np.float32)
        p['bn_sm/BatchNorm/moving_mean'] = rs.uniform(0.0, 1e-3, 10).astype(np.float32)
        p['bn_sm/BatchNorm/moving_variance'] = rs.uniform(0.5, 1.5, 10).astype(np.float32)
    return p


def synthetic_priors(seed=5, n_joints=N_JOINTS):
    """Seeded stand-ins for the 120x180 displacement histograms: a few Gaussian bumps per
    pair, normalised to sum 1 like the real ones (prepare_pairwise_distribution.py:46)."""
    rs = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:120, 0:180].astype(np.float64)
    out = {}
    for key in pair_keys(n_joints):
        pd = np.zeros((120, 180))
        for _ in range(3):
            cy, cx = 60 + rs.normal(0, 12), 90 + rs.normal(0, 18)
            sy, sx = rs.uniform(2, 8), rs.uniform(2, 10)
            pd += rs.uniform(0.2, 1.0) * np.exp(-0.5 * (((yy - cy) / sy) ** 2 + ((xx - cx) / sx) ** 2))
        out[key] = pd / pd.sum()
    return out


def make_images(batch, seed=1234, height=480, width=720):
    """U[0,1) float32 NHWC images (data.py:129-130 scales JPEG bytes to [0,1])."""
    return np.random.RandomState(seed).random_sample((batch, height, width, 3)).astype(np.float32)


def make_torso(batch, seed=4321, hm_height=60, hm_width=90):
    """[B,60,90,1] torso heat maps: 3x3 binomial blob at a uniform interior cell
    (data.py:112-114,180-186)."""
    rs = np.random.RandomState(seed)
    kern = np.outer([1, 2, 1], [1, 2, 1]).astype(np.float32) / 16
    t = np.zeros((batch, hm_height, hm_width, 1), np.float32)
    for b in range(batch):
        r, c = rs.randint(1, hm_height - 1), rs.randint(1, hm_width - 1)
        t[b, r - 1:r + 2, c - 1:c + 2, 0] = kern
    return t


def make_targets(batch, seed=4321, hm_height=60, hm_width=90, n_channels=10):
    """[B,60,90,10] target heat maps y_in (main.py:488): one 3x3 binomial blob per joint and image
    (data.py:112-114,180-186).  Channel 9 (torso) uses the same stream as `make_torso(batch, seed)`
    so `make_targets(...)[..., 9:]` equals it."""
    kern = np.outer([1, 2, 1], [1, 2, 1]).astype(np.float32) / 16
    t = np.zeros((batch, hm_height, hm_width, n_channels), np.float32)
    t[..., n_channels - 1:] = make_torso(batch, seed, hm_height, hm_width)
    rs = np.random.RandomState(seed + 1)
    for b in range(batch):
        for k in range(n_channels - 1):
            r, c = rs.randint(1, hm_height - 1), rs.randint(1, hm_width - 1)
            t[b, r - 1:r + 2, c - 1:c + 2, k] = kern
    return t
