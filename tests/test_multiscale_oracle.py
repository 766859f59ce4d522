"""Known-answer tests of the restated skimage.resize semantics and the scale windows of the
multi-scale evaluation (main.py:326-379); two formulations must agree."""
import numpy as np
from hypothesis import given, settings, strategies as st

from oracle import multiscale_oracle as M


def test_scale_windows_match_reference_arithmetic():
    w = M.scale_windows(M.PAD_ARRAY, M.CROP_ARRAY, 480, 720)
    assert w[0] == (-24, -36, 528, 792)            # pad 1.1: round(480*0.1/2)=24, round(720*0.1/2)=36
    assert w[3] == (-96, -144, 672, 1008)          # pad 1.4
    assert w[4] == (72, 108, 336, 504)             # crop 0.7: h1=round(0.15*480)=72, h2-h1=round(0.7*480)=336
    assert w[7] == (0, 0, 480, 720)                # crop 1.0 = identity window
    b = M.back_windows(M.PAD_ARRAY, M.CROP_ARRAY, 60, 90)
    assert b[0] == (3, 4, 55, 82)                  # 1/1.1: round(2.727)=3, round(4.09)=4, round(54.5)=55 (banker's: 54.545->55), round(81.8)=82
    assert b[4] == (-13, -19, 86, 128)             # 1/0.7: round(12.857)=13, round(19.29)=19
    assert b[7] == (0, 0, 60, 90)


def test_resize_identity_and_constant_border():
    x = np.random.RandomState(0).random_sample((6, 9, 2)) + 0.5       # all > 0: cval outside [min,max]
    np.testing.assert_allclose(M.resize_skimage(x, 6, 9), x, rtol=1e-14)          # same size = identity
    up = M.resize_skimage(x, 12, 18)
    # src = 0.5*(dst+0.5)-0.5: dst 0 -> -0.25 blends the (zero) outside with pixel 0 -> 0.75*x, then clipped up to min
    assert np.isclose(up[0, 0, 0], max(0.75 * 0.75 * x[0, 0, 0], x.min()))
    assert up.min() >= x.min() - 1e-15 and up.max() <= x.max() + 1e-15                # clip=True
    assert np.isclose(up[1, 1, 0], (0.75 * (0.75 * x[0, 0, 0] + 0.25 * x[0, 1, 0]) + 0.25 * (0.75 * x[1, 0, 0] + 0.25 * x[1, 1, 0])))


def test_take_window_pads_with_zeros_and_crops():
    x = np.arange(2 * 3 * 1, dtype=float).reshape(2, 3, 1) + 1
    p = M.take_window(x, (-1, -2, 4, 7))
    assert p.shape == (4, 7, 1) and p[0].sum() == 0 and p[:, :2].sum() == 0
    np.testing.assert_array_equal(p[1:3, 2:5], x)
    np.testing.assert_array_equal(M.take_window(x, (1, 1, 1, 2)), x[1:2, 1:3])


@settings(max_examples=30, deadline=None)
@given(h=st.integers(2, 20), w=st.integers(2, 20), oh=st.integers(1, 30), ow=st.integers(1, 30),
       lo=st.sampled_from([0.0, 0.3, -0.5]), seed=st.integers(0, 10 ** 6))
def test_two_formulations_agree(h, w, oh, ow, lo, seed):
    x = np.random.RandomState(seed).random_sample((h, w, 3)) + lo
    np.testing.assert_allclose(M.resize_skimage(x, oh, ow), M.resize_skimage_scipy(x, oh, ow), rtol=1e-12, atol=1e-12)


def test_pipeline_shapes_and_identity_scale():
    rs = np.random.RandomState(3)
    x = rs.random_sample((480, 720, 3)).astype(np.float32)
    xs = M.get_different_scales(x)
    assert xs.shape == (8, 480, 720, 3)
    np.testing.assert_allclose(xs[7], x, rtol=1e-7)                      # crop 1.0 is the image itself
    assert xs[0][:20].max() == 0 and xs[0][:, :30].max() == 0          # pad 1.1: a zero frame around the shrunk image
    hm = rs.random_sample((8, 60, 90, 9))
    back = M.scale_hm_back(hm)
    assert back.shape == (8, 60, 90, 9)
    np.testing.assert_allclose(back[7], hm[7], rtol=1e-12)
    c = M.argmax_hm(np.expand_dims(np.average(back, axis=0), 0))
    assert c.shape == (2, 9) and c[0].max() < 60 and c[1].max() < 90
