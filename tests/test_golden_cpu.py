"""The oracle against the committed golden vectors, on CPU.  The goldens were written by the
NumPy formulation in float64; here the *other* formulation (torch/SciPy) must reproduce them,
and the priors must reproduce what the reference's own prepare_pairwise_distribution.py emits
(that check runs where /root/reference exists; elsewhere the recorded properties are checked)."""
import os

import numpy as np
import pytest
import torch

from golden_util import batch_golden, config_batch, flic_priors, full_inputs, load, seeds, stats
from joint_cnn_mrf_amd import synth
from oracle import jcm_oracle as O
from oracle import jcm_oracle_torch as T


@pytest.fixture(scope='module')
def pri():
    return flic_priors()


def test_flic_priors_properties(pri):
    """90 pairs, each a smoothed histogram summing to 1 (prepare_pairwise_distribution.py:44-45)."""
    assert len(pri) == 90
    for k, v in pri.items():
        assert v.shape == (120, 180) and v.dtype == np.float64
        assert abs(v.sum() - 1.0) < 1e-9 and v.min() >= 0
    assert max(v.max() for v in pri.values()) < 0.05
    assert np.unravel_index(pri['nose_torso'].argmax(), (120, 180)) == (49, 89)   # nose sits ~11 cells above the torso centre
    cells = load('flic_train_cells')
    assert cells.shape == (3987, 10, 2) and cells[:, :, 0].max() <= 59 and cells[:, :, 1].max() <= 89


@pytest.mark.skipif(not os.path.exists('/root/reference/prepare_pairwise_distribution.py'), reason='reference tree absent')
def test_prior_smoothing_kernel_matches_reference_source():
    """The 9x9 binomial kernel is the one the reference builds (its lines 13-14)."""
    src = open('/root/reference/prepare_pairwise_distribution.py').read()
    assert 'np.array([[1, 8, 28, 56, 70, 56, 28, 8, 1]], dtype=np.uint16) / 256' in src
    from joint_cnn_mrf_amd import priors
    assert priors.SMOOTH_KERNEL.shape == (9, 9) and abs(priors.SMOOTH_KERNEL.sum() - 1.0) < 1e-12


def test_conv_mrf_golden_both_formulations():
    prior, lik = load('conv_mrf_prior').astype(np.float64), load('conv_mrf_lik').astype(np.float64)
    pre, post = load('conv_mrf_pre'), load('conv_mrf_post')
    assert pre.shape == (2, 61, 91, 1) and post.shape == (2, 60, 90, 1)
    np.testing.assert_allclose(O.conv_mrf_pre(prior, lik), pre, rtol=2e-6)
    np.testing.assert_allclose(T.conv_mrf(prior, lik), post, rtol=2e-6)
    # KAT2 on real data: output row 0 is pre-resize row 0; the resize is not a crop
    np.testing.assert_allclose(post[:, 0, 0], pre[:, 0, 0], rtol=1e-6)
    assert np.abs(post[:, 59, 89] - pre[:, 59, 89]).max() > 0


def test_full_size_spatial_model_golden(pri):
    """Second formulation reproduces the stored SM logits/coords from the stored PD logits."""
    x, torso, _ = full_inputs()
    pd_prob = O.spatial_softmax(load('full_pd_logits').astype(np.float64))
    hm10 = np.concatenate([pd_prob, torso.astype(np.float64)], axis=3)
    for kind in ('init', 'trained'):
        sp = synth.make_sm_params(pri, kind=kind, seed=seeds()['sm'])
        got = T.spatial_model(hm10, sp)
        ref = load('full_sm_logits_' + kind)
        np.testing.assert_allclose(got, ref, atol=5e-5 * max(1.0, np.abs(ref).max()), rtol=0)
        np.testing.assert_array_equal(T.argmax_coords(T.spatial_softmax(got)), load('full_sm_coords_' + kind))


def test_full_size_part_detector_golden():
    """torch formulation (float32, oneDNN) vs the float64 NumPy golden at FULL size: also
    sizes the fp32 noise floor the GPU tolerance is set against."""
    x, _, p = full_inputs()
    ref = load('full_pd_logits')
    got = T.model(x[:1], p, dtype=torch.float32)
    err = np.abs(got - ref[:1]).max()
    assert err < 2e-4 * max(1.0, np.abs(ref).max()), err
    np.testing.assert_array_equal(O.argmax_coords(O.spatial_softmax(got.astype(np.float64))), load('full_pd_coords')[:1])
    st = stats()
    assert st['pd_top2_margin'] > 1e-2 and st['sm_top2_margin_trained'] > 2e-3


def test_batch_goldens_second_formulation(pri):
    """tests/golden/batch64.npz / batch256.npz (the value fixtures of the configs[1] / configs[2] batches): the torch formulation in float32
    reproduces the stored float64 logits of the LAST image of the 64-image batch (rebuilt from seeds by golden_util.config_batch), the stored
    coordinates are the arg-max of the stored logits, and the spatial model's second formulation reproduces the stored SM logits."""
    g = batch_golden(64)
    assert g['idx'].tolist() == [7, 8, 16, 31, 32, 47, 62, 63] and g['pd_logits'].shape == (8, 60, 90, 9)
    x, torso = config_batch(64)
    _, _, p = full_inputs()
    got = T.model(x[63:64], p, dtype=torch.float32)
    ref = g['pd_logits'][7:8]
    assert np.abs(got - ref).max() < 2e-4 * max(1.0, np.abs(ref).max())
    for B in (64, 256):
        g = batch_golden(B)
        for st in ('pd', 'sm'):
            lg = g[st + '_logits'].astype(np.float64)
            np.testing.assert_array_equal(O.argmax_coords(O.spatial_softmax(lg)), g[st + '_coords'])
    g = batch_golden(64)
    sp = synth.make_sm_params(pri, kind='trained', seed=seeds()['sm'])
    hm10 = np.concatenate([O.spatial_softmax(g['pd_logits'][6:8].astype(np.float64)), torso[62:64].astype(np.float64)], axis=3)
    sm = T.spatial_model(hm10, sp)
    np.testing.assert_allclose(sm, g['sm_logits'][6:8], atol=5e-5 * max(1.0, np.abs(g['sm_logits']).max()), rtol=0)
