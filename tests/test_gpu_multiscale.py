"""Multi-scale test-time evaluation (main.py:326-425) on the GPU against the NumPy oracle."""
import numpy as np
import pytest
import torch

import joint_cnn_mrf_amd  # noqa: F401
from joint_cnn_mrf_amd import multiscale as MS
from joint_cnn_mrf_amd import synth
from oracle import jcm_oracle as O
from oracle import multiscale_oracle as MO

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32), device='cuda:0')


@pytest.fixture(scope='module')
def eng():
    from joint_cnn_mrf_amd.engine import Engine
    p = synth.make_pd_params(debug=True, bn='trained', conv6_gain=8.0)
    p.update(synth.make_sm_params(synth.synthetic_priors(), kind='trained'))
    e = Engine(device=0).load_params(p)
    e._params = p
    yield e
    e.close()


def test_get_different_scales_and_back(eng):
    rs = np.random.RandomState(5)
    x = rs.random_sample((2, 480, 720, 3)).astype(np.float32) * 0.8 + 0.1       # strictly positive: exercises the clip
    got = MS.get_different_scales(eng, dev(x), MS.PAD_ARRAY, MS.CROP_ARRAY, 480, 720).cpu().numpy()
    assert got.shape == (16, 480, 720, 3)
    for i in range(2):
        ref = MO.get_different_scales(x[i])
        np.testing.assert_allclose(got[8 * i:8 * i + 8], ref, atol=2e-7, rtol=0)
    hm = O.spatial_softmax(rs.standard_normal((16, 60, 90, 9)) * 3).astype(np.float32)
    back = MS.scale_hm_back(eng, dev(hm), MS.PAD_ARRAY, MS.CROP_ARRAY, 60, 90).cpu().numpy()
    for i in range(2):
        np.testing.assert_allclose(back[8 * i:8 * i + 8], MO.scale_hm_back(hm[8 * i:8 * i + 8]), atol=1e-9, rtol=1e-6)
    mean = eng.group_mean(dev(back), 8).cpu().numpy()
    np.testing.assert_allclose(mean, back.reshape(2, 8, 60, 90, 9).mean(axis=1, dtype=np.float64), rtol=1e-6, atol=1e-12)


def test_get_predictions_matches_oracle(eng):
    """Whole caller: rescale -> tower (debug-size network) -> scale back -> average -> arg-max."""
    p = eng._params
    X = synth.make_images(2, seed=61)
    Y = np.concatenate([np.zeros((2, 60, 90, 9), np.float32), synth.make_torso(2, seed=62)], axis=3)
    pd, sm = MS.get_predictions(eng, X, Y, use_sm=True, images_per_forward=2)
    assert pd.shape == sm.shape == (2, 9, 2)

    def forward(xs, ys):
        r = O.forward(xs, ys[:, :, :, 9:], p)
        return r['pd_prob'], r['sm_prob']

    # the same pipeline step by step, to compare the averaged maps themselves
    xs = MS.get_different_scales(eng, dev(X), MS.PAD_ARRAY, MS.CROP_ARRAY, 480, 720)
    torso = dev(Y[:, :, :, 9:]).repeat_interleave(8, dim=0).contiguous()
    r = eng.forward(xs, torso, use_sm=True)
    g_pd = eng.group_mean(MS.scale_hm_back(eng, r['pd_prob'], MS.PAD_ARRAY, MS.CROP_ARRAY, 60, 90), 8).cpu().numpy()
    g_sm = eng.group_mean(MS.scale_hm_back(eng, r['sm_prob'], MS.PAD_ARRAY, MS.CROP_ARRAY, 60, 90), 8).cpu().numpy()
    for i in (1,):      # the second image of the group (the oracle's 8 CPU forwards per image are what this test spends its time on)
        c_pd, c_sm, hm_pd, hm_sm = MO.predict_one(X[i], Y[i], forward)
        for got_c, ref_c, got_hm, ref_hm in ((pd[:, :, i], c_pd, g_pd[i], hm_pd[0]), (sm[:, :, i], c_sm, g_sm[i], hm_sm[0])):
            err = np.abs(got_hm - ref_hm).max()
            assert err <= 1e-4 and err <= 2e-3 * ref_hm.max()                 # heat-map bar, and a relative one
            flat = np.sort(ref_hm.reshape(5400, 9), axis=0)
            clear = (flat[-1] - flat[-2]) > 4 * err                          # arg-max is well posed there
            np.testing.assert_array_equal(got_c[:, clear], ref_c[:, clear])
            assert clear.sum() >= 5
