"""Parameter-file format (TF variable naming) and Hypothesis property tests between the two
oracle formulations on small random shapes."""
import os

import numpy as np
import pytest
import torch
from hypothesis import given, settings, strategies as st

import joint_cnn_mrf_amd  # noqa: F401
from joint_cnn_mrf_amd import checkpoint, synth
from oracle import jcm_oracle as O
from oracle import jcm_oracle_torch as T


def test_npz_roundtrip_and_validation(tmp_path):
    p = synth.make_pd_params(debug=True)
    p.update(synth.make_sm_params(synth.synthetic_priors()))
    p['n_iters'] = np.zeros((), np.float32)                    # saved by the reference, not read by inference
    p['conv5/weights/Adam'] = np.zeros((9, 9, 128, 128), np.float32)
    path = str(tmp_path / 'model.npz')
    checkpoint.save_npz(path, p)
    q = checkpoint.load_npz(path, debug=True)
    want = checkpoint.expected_shapes(debug=True)
    assert sorted(q) == sorted(want) and len(want) == 6 * 13 + 2 + 4 + 162
    for k in want:
        np.testing.assert_array_equal(q[k], p[k])
    assert want['conv1_halfres/weights'] == (5, 5, 3, 16) and want['energy_lsho_torso'] == (1, 120, 180, 1)
    bad = dict(p)
    del bad['conv6/biases']
    bad['bias_nose_lwri'] = np.zeros((60, 90), np.float32)
    checkpoint.save_npz(path, bad)
    with pytest.raises(ValueError) as ei:
        checkpoint.load_npz(path, debug=True)
    assert 'missing conv6/biases' in str(ei.value) and 'bias_nose_lwri' in str(ei.value)
    with pytest.raises(ValueError):
        checkpoint.load_npz(path, debug=False)                  # full-size model expects 64..512 filters


@settings(max_examples=25, deadline=None)
@given(h=st.integers(5, 23), w=st.integers(5, 23), cin=st.integers(1, 4), cout=st.integers(1, 4),
       k=st.sampled_from([5, 9]), stride=st.sampled_from([1, 2]), seed=st.integers(0, 10 ** 6))
def test_conv_same_padding_property(h, w, cin, cout, k, stride, seed):
    """NumPy slicing conv == torch conv with explicit asymmetric padding, any size / stride."""
    rs = np.random.RandomState(seed)
    x = rs.standard_normal((2, h, w, cin))
    wt = rs.standard_normal((k, k, cin, cout))
    a = O.conv2d_same(x, wt, stride)
    b = T.conv2d_same(torch.as_tensor(x).permute(0, 3, 1, 2), torch.as_tensor(wt), stride).permute(0, 2, 3, 1).numpy()
    assert a.shape == (2, -(-h // stride), -(-w // stride), cout)
    np.testing.assert_allclose(a, b, rtol=1e-10, atol=1e-10)


@settings(max_examples=25, deadline=None)
@given(h=st.integers(1, 31), w=st.integers(1, 31), oh=st.integers(1, 40), ow=st.integers(1, 40), seed=st.integers(0, 10 ** 6))
def test_resize_property(h, w, oh, ow, seed):
    x = np.random.RandomState(seed).standard_normal((1, h, w, 2))
    a = O.resize_bilinear_tf1(x, oh, ow)
    b = T.resize_bilinear_tf1(torch.as_tensor(x).permute(0, 3, 1, 2), oh, ow).permute(0, 2, 3, 1).numpy()
    np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-12)
    assert a.min() >= x.min() - 1e-12 and a.max() <= x.max() + 1e-12      # bilinear is a convex combination


@settings(max_examples=20, deadline=None)
@given(h=st.integers(1, 17), w=st.integers(1, 17), seed=st.integers(0, 10 ** 6))
def test_pool_property(h, w, seed):
    x = np.random.RandomState(seed).standard_normal((1, h, w, 3))
    a = O.max_pool_same(x)
    b = T.max_pool_same(torch.as_tensor(x).permute(0, 3, 1, 2)).permute(0, 2, 3, 1).numpy()
    assert a.shape == (1, (h + 1) // 2, (w + 1) // 2, 3)
    np.testing.assert_array_equal(a, b)


# ---------------------------------------------------------------- data preparation
def test_flic_target_heat_maps_from_annotations():
    """data.py:163-189 on the real FLIC annotations (tests/golden/flic_train_xy.npy): the arg-max cells of the
    rebuilt y_train equal the committed cells the prior builder was pinned with; make_golden.py asserts the
    full arrays (train and test) bit for bit against the reference script's logic."""
    import os
    from joint_cnn_mrf_amd import data
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    xy = np.load(os.path.join(here, 'flic_train_xy.npy'))
    gold = np.load(os.path.join(here, 'flic_train_cells.npy'))
    y = data.target_heat_maps(data.joint_cells(xy[:600]))
    assert y.shape == (600, 60, 90, 10) and y.dtype == np.float32
    flat = y.reshape(600, 5400, 10)
    idx = flat.argmax(axis=1)
    np.testing.assert_array_equal(np.stack([idx // 90, idx % 90], axis=2), gold[:600])
    assert abs(float(flat.max()) - 0.25) < 1e-7                     # centre of the [1,2,1]x[1,2,1]/16 blob
    s = flat.sum(axis=1)
    assert (s <= 1 + 1e-6).all() and (s[:, 9] > 0.99).mean() > 0.95  # blobs cut only at the image border


def test_flic_backward_pose_flip_and_border_clamp():
    from joint_cnn_mrf_amd import data
    xy = np.zeros((2, 2, 9))
    xy[:, 0] = [100, 110, 120, 300, 310, 320, 150, 250, 200]         # x: lsho lelb lwri rsho relb rwri lhip rhip nose
    xy[:, 1] = [80, 120, 160, 80, 120, 160, 240, 240, 40]
    xy[1, 0, 6], xy[1, 0, 7] = 250, 150                              # image 1: frontal (left hip right of the right hip)
    xy[1, :, 8] = [900, 600]                                         # nose annotated outside the image
    c = data.joint_cells(xy)
    # image 0 (lhip.x < rhip.x): the reference's view-based swap leaves left := right (data.py:35-49)
    np.testing.assert_array_equal(c[0, 0], c[0, 3])
    np.testing.assert_array_equal(c[0, 2], c[0, 5])
    np.testing.assert_array_equal(c[0, 6], c[0, 7])
    np.testing.assert_array_equal(c[1, 0], [10, 12])                 # untouched: (80/8, 100/8)
    np.testing.assert_array_equal(c[1, 8], [60, 90])                 # clamped to (480, 720) / 8: blob cut by the border
    y = data.target_heat_maps(c)
    assert y[1, 59, 89, 8] == np.float32(1 / 16) and y[1, :, :, 8].sum() == np.float32(1 / 16)


def test_piecewise_learning_rate_schedule():
    """lr_tf (main.py:467-469,492): boundaries round(0.7/0.8/0.9 * total), values lr, lr/2, lr/5, lr/10; the host helper
    agrees with the restatement on every update of a short and a long run, including the boundary updates."""
    from joint_cnn_mrf_amd.train import piecewise_lr
    from oracle.train_oracle import piecewise_lr as ref
    for total, lr in ((10, 0.001), (8541, 0.01), (3, 1.0)):
        for n in range(total + 2):
            assert piecewise_lr(n, total, lr) == ref(n, total, lr)
    assert piecewise_lr(0, 100, 1.0) == 1.0 and piecewise_lr(70, 100, 1.0) == 1.0 and piecewise_lr(71, 100, 1.0) == 0.5
    assert piecewise_lr(81, 100, 1.0) == 0.2 and piecewise_lr(91, 100, 1.0) == 0.1 and piecewise_lr(1000, 100, 1.0) == 0.1


# ---------------------------------------------------------------------------------------------------- tf.train.Saver files
def test_crc32c_known_answers():
    """RFC 3720 B.4 vectors + the leveldb mask (crc32c::Mask / Unmask)."""
    from joint_cnn_mrf_amd import tf_checkpoint as C
    assert C.crc32c(bytes(32)) == 0x8a9136aa
    assert C.crc32c(b'\xff' * 32) == 0x62a8ab43
    assert C.crc32c(bytes(range(32))) == 0x46dd794e
    assert C.crc32c(b'123456789') == 0xe3069283
    big = np.random.RandomState(0).bytes(70001)             # long enough for the libjcm (hardware) route
    c = 0
    for b in big:                                           # byte-at-a-time table reference
        c = C.crc32c(bytes([b]), c)
    assert C.crc32c(big) == c == C.crc32c(big[30000:], C.crc32c(big[:30000]))
    assert C.unmask_crc(C.mask_crc(0xdeadbeef)) == 0xdeadbeef and C.mask_crc(0xdeadbeef) != 0xdeadbeef


def test_tf_checkpoint_v2_round_trip(tmp_path):
    """Writer and reader of the tf.train.Saver checkpoint-V2 files (main.py:604,612,666) against each other: model
    variables under the reference's names, Adam slots, scalars (n_iters int32, beta powers), enough variables for several
    restart groups, a block size small enough for several data blocks."""
    from joint_cnn_mrf_amd import synth
    from joint_cnn_mrf_amd import tf_checkpoint as C
    p = synth.make_pd_params(debug=True, bn='trained')
    p.update(synth.make_sm_params(synth.synthetic_priors(), kind='trained'))
    state = dict(p)
    state['conv5/weights/Adam'] = np.full_like(p['conv5/weights'], 0.25)
    state['conv5/weights/Adam_1'] = np.full_like(p['conv5/weights'], 1e-9)
    state['beta1_power'], state['beta2_power'], state['n_iters'] = np.float32(0.9 ** 5), np.float32(0.999 ** 5), np.int32(5)
    prefix = str(tmp_path / 'models_ex' / 'run_lr=0.001-17')
    C.save_checkpoint(prefix, state)
    assert sorted(f.name for f in (tmp_path / 'models_ex').iterdir()) == ['run_lr=0.001-17.data-00000-of-00001', 'run_lr=0.001-17.index']
    back = C.load_checkpoint(prefix)
    assert set(back) == set(state)
    for k, v in state.items():
        assert back[k].shape == np.shape(v) and back[k].dtype == np.asarray(v).dtype, k
        np.testing.assert_array_equal(back[k], v, err_msg=k)
    assert back['n_iters'].dtype == np.int32 and back['n_iters'].shape == ()
    listed = {n: (s, d) for n, s, d in C.list_variables(prefix)}
    assert listed['conv6/weights'] == ((9, 9, 128, 9), np.dtype('float32')) and listed['energy_lsho_lelb'][0] == (1, 120, 180, 1)
    # data file: tensors back to back in key order
    assert os.path.getsize(prefix + '.data-00000-of-00001') == sum(np.asarray(v).nbytes for v in state.values())
    # table structure: magic number at the end, header entry first, keys sorted, several data blocks with a tiny block size
    small = str(tmp_path / 'small.index')
    items = [(('k%04d' % i).encode(), bytes([i % 256]) * (i % 37)) for i in range(500)]
    C.write_table(small, items, block_size=256)
    assert C.read_table(small) == items
    raw = open(small, 'rb').read()
    assert raw[-8:] == bytes.fromhex('57fb808b247547db')            # kTableMagicNumber 0xdb4775248b80fb57, little endian
    assert C.read_table(prefix + '.index')[0] == (b'', b'\x08\x01\x1a\x02\x08\x01')   # num_shards 1, version.producer 1
    # corruption is detected: flip one byte of a tensor / of a table block
    with open(prefix + '.data-00000-of-00001', 'r+b') as fh:
        fh.seek(100)
        b0 = fh.read(1)
        fh.seek(100)
        fh.write(bytes([b0[0] ^ 1]))
    with pytest.raises(ValueError, match='CRC-32C'):
        C.load_checkpoint(prefix)
    bad = bytearray(raw)
    bad[10] ^= 0x40
    open(small, 'wb').write(bytes(bad))
    with pytest.raises(ValueError, match='checksum'):
        C.read_table(small)


def test_snappy_blocks_are_readable():
    """Tables written with the leveldb default (snappy, block type 1) must be readable too: hand-assembled stream with a
    literal and the three copy forms."""
    from joint_cnn_mrf_amd import tf_checkpoint as C
    #  length 18 | literal "abcd" | copy1 off 4 len 4 | copy2 off 8 len 6 | literal "xyzw"
    stream = bytes([18]) + bytes([(4 - 1) << 2]) + b'abcd' + bytes([((4 - 4) << 2) | 1 | (0 << 5), 4]) + bytes([((6 - 1) << 2) | 2, 8, 0]) + bytes([(4 - 1) << 2]) + b'xyzw'
    assert C._snappy_uncompress(stream) == b'abcdabcdabcdabxyzw'


def test_data_prepare_writes_the_four_arrays_from_frames(tmp_path):
    """data.prepare end to end, image half included (data.py:120-196): an annotation file in data_FLIC.mat's structure
    (examples[0][i][2] = [2,29] coordinates, [3] = file name, [7] = is-train flag) plus JPEG frames on disk ->
    x_{train,test}_flic.npy (fp32 RGB / 255, [N,480,720,3]) and y_{train,test}_flic.npy ([N,60,90,10]), split and
    ordered as the reference's loops."""
    import scipy.io
    from PIL import Image
    from joint_cnn_mrf_amd import data
    rs = np.random.RandomState(12)
    frames = tmp_path / 'images_FLIC'
    frames.mkdir()
    n = 5
    ex = np.zeros((1, n), dtype=[('poselet_hit_idx', 'O'), ('moviename', 'O'), ('coords', 'O'), ('filepath', 'O'), ('imgdims', 'O'),
                                 ('currframe', 'O'), ('torsobox', 'O'), ('istrain', 'O'), ('istest', 'O'), ('isbad', 'O'), ('isunchecked', 'O')])
    is_train = [1, 0, 1, 1, 0]
    imgs = []
    for i in range(n):
        xy = np.full((2, 29), np.nan)
        xy[0, :] = rs.uniform(40, 680, 29)
        xy[1, :] = rs.uniform(40, 440, 29)
        name = 'movie-%08d.jpg' % i
        img = (rs.random_sample((480, 720, 3)) * 255).astype(np.uint8)
        Image.fromarray(img).save(str(frames / name), quality=95)
        imgs.append(np.asarray(Image.open(str(frames / name)).convert('RGB'), np.float32) / 255)
        ex[0, i]['coords'], ex[0, i]['filepath'], ex[0, i]['istrain'] = xy, np.array([name]), np.array([[is_train[i]]])
    mat = str(tmp_path / 'data_FLIC.mat')
    scipy.io.savemat(mat, {'examples': ex})
    out = data.prepare(mat, images_dir=str(frames), out_dir=str(tmp_path))
    assert out == {'y_train': (3, 60, 90, 10), 'x_train': (3, 480, 720, 3), 'y_test': (2, 60, 90, 10), 'x_test': (2, 480, 720, 3)}
    x_train, x_test = np.load(tmp_path / 'x_train_flic.npy'), np.load(tmp_path / 'x_test_flic.npy')
    assert x_train.dtype == np.float32 and 0 <= x_train.min() and x_train.max() <= 1
    np.testing.assert_array_equal(x_train, np.stack([imgs[0], imgs[2], imgs[3]]))
    np.testing.assert_array_equal(x_test, np.stack([imgs[1], imgs[4]]))
    y_train = np.load(tmp_path / 'y_train_flic.npy')
    xy_all, _names, tr = data.load_flic(mat)
    np.testing.assert_array_equal(y_train, data.target_heat_maps(data.joint_cells(xy_all[tr])))
    np.testing.assert_allclose(y_train.sum(axis=(1, 2)), 1.0, rtol=1e-6)      # none of these blobs touches the border
    # the files are exactly what main.get_dataset reads
    from joint_cnn_mrf_amd import main as M
    xs = M.get_dataset(str(tmp_path))
    assert [a.shape[0] for a in xs] == [3, 3, 2, 2]
