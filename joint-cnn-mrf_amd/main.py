"""Host-side mirror of the reference's main.py for the joint-heat-map inference path.

Same function names, argument order and NHWC layouts as /root/reference/main.py, so a
caller of `model`, `conv_mrf`, `spatial_model`, `spatial_softmax` (graph at main.py:522-531)
switches by importing this module; tensors are torch-ROCm instead of tf.Tensor, and the
arithmetic runs in libjcm's HIP kernels (no TensorFlow, no CPU fallback).

Where the reference reads module globals (`hps`, `flag_train`, `pairwise_energies`,
`pairwise_biases`, `joint_names`, `joint_dependence`, `n_joints`, `hm_height`, `hm_width`)
this module keeps the same names; they are filled by `configure()` instead of at import
(the reference parses argv and loads `.npy` files at import time, main.py:440,459).

CLI (the reference's flags, main.py:428-439): `python -m joint_cnn_mrf_amd.main --gpus 0 1 --use_sm --batch_size 64`
evaluates the test split (single scale sharded over the listed devices, or `--multiscale`); `--train` runs the
reference's epoch loop (main.py:620-667) with one tower per device, eval_error after every epoch and tf.train.Saver
checkpoints; `--restore --restore_path P` resumes from one.  Data: the .npy files data.py prepares, or `--synthetic`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

from . import synth
from .engine import Engine

# main.py:18-26
joint_names = np.array(['lsho', 'lelb', 'lwri', 'rsho', 'relb', 'rwri', 'lhip', 'rhip', 'nose', 'torso'])
joint_dependence = {}
for _joint in joint_names:
    joint_dependence[_joint] = [_c for _c in joint_names if _c != _joint]

n_joints = 9                      # main.py:458
in_height, in_width = 480, 720    # data.py:119
hm_height, hm_width = 60, 90      # data.py:180
flag_train = False                # inference only: BatchNorm uses moving statistics (main.py:406)

hps = argparse.Namespace(debug=False, train=False, gpus=[0], restore=False, use_sm=True, data_augm=False,
                         batch_size=14)            # defaults of main.py:428-439
pairwise_energies, pairwise_biases = {}, {}        # '<j>_<c>' -> tensor, main.py:478-487
_engine = None


def build_parser():
    """The reference's eleven flags (main.py:428-439), same names, types and defaults (but --gpus, whose reference default
    [6] names a device of the authors' server), plus what the reference hard-codes in module globals: where the data and
    the checkpoints live, and an explicit switch for generated data."""
    parser = argparse.ArgumentParser(description='Define hyperparameters.')
    parser.add_argument('--debug', action='store_true', help='True if we want to debug.')
    parser.add_argument('--train', action='store_true', help='True if we want to train the model.')
    parser.add_argument('--gpus', nargs='+', type=int, default=[0], help='GPU indices.')
    parser.add_argument('--restore', action='store_true', help='True if we want to restore the model.')
    parser.add_argument('--use_sm', action='store_true', help='True if we want to use the Spatial Model.')
    parser.add_argument('--data_augm', action='store_true', help='True if we want to use data augmentation.')
    parser.add_argument('--n_epochs', type=int, default=30, help='Number of epochs.')
    parser.add_argument('--batch_size', type=int, default=14, help='Batch size.')
    parser.add_argument('--optimizer', type=str, default='adam', help='momentum or adam')
    parser.add_argument('--lr', type=float, default=0.001, help='Learning rate.')
    parser.add_argument('--lmbd', type=float, default=0.001, help='Regularization coefficient.')
    # not in the reference (module globals / hard-coded paths there)
    parser.add_argument('--data_dir', default='.', help='directory of x_*_flic.npy, y_*_flic.npy, pairwise_distribution.pickle (main.py:290-298).')
    parser.add_argument('--synthetic', action='store_true', help='generated images / targets / priors instead of the FLIC files.')
    parser.add_argument('--synthetic_size', type=int, default=56, help='number of generated training examples with --synthetic.')
    parser.add_argument('--model_path', default=model_path, help='checkpoint directory (main.py:444).')
    parser.add_argument('--restore_path', default=None, help='checkpoint prefix (tf.train.Saver files) or .npz for --restore (main.py:443,612).')
    parser.add_argument('--precision', default='fp32', choices=['fp32', 'bf16'], help='arithmetic of the kernels.')
    parser.add_argument('--multiscale', action='store_true', help='evaluation run: 8-scale test-time evaluation (get_predictions, main.py:382-425).')
    parser.add_argument('--predictions', default=None, help='evaluation run: write flic_pred_pd / flic_pred_sm to this .mat file (main.py:675).')
    parser.add_argument('--seed', type=int, default=0, help='shuffling seed.')
    return parser


def configure(params, device=0, precision='fp32', debug=None):
    """Create the engine for `device` and load `params` (dict keyed by TF variable names).
    Replaces graph construction + tf.Session + Saver.restore (main.py:474-487,606-612)."""
    global _engine, pairwise_energies, pairwise_biases
    if debug is not None:
        hps.debug = bool(debug)
    if _engine is not None:
        _engine.close()
    _engine = Engine(device=device, precision=precision, n_joints=n_joints).load_params(params)
    dev = _engine.device
    pairwise_energies = {k[len('energy_'):]: torch.as_tensor(np.asarray(v), device=dev) for k, v in params.items() if k.startswith('energy_')}
    pairwise_biases = {k[len('bias_'):]: torch.as_tensor(np.asarray(v), device=dev) for k, v in params.items() if k.startswith('bias_')}
    return _engine


def engine():
    if _engine is None:
        raise RuntimeError('call joint_cnn_mrf_amd.main.configure(params) first (replaces sess.run setup)')
    return _engine


# ------------------------------------------------------------------ layer wrappers (main.py:128-181)
def conv_layer(x, size, stride, n_in, n_out, name, last_layer=False):
    """main.py:156-169."""
    if x.shape[-1] != n_in:
        raise ValueError('conv_layer %s: input has %d channels, n_in=%d' % (name, x.shape[-1], n_in))
    return engine().conv_layer(x, name, stride, last_layer=last_layer, n_out=n_out)


def max_pool_layer(x, size, stride):
    """main.py:172-174 (the model only uses 2x2/2)."""
    if (size, stride) != (2, 2):
        raise ValueError('only the 2x2 stride-2 SAME pool of the reference model is implemented')
    return engine().max_pool(x)


def resize_images(x, size):
    """tf.image.resize_images(x, [h, w]) with TF-1.x defaults (main.py:51,58,60,67)."""
    return engine().resize_bilinear(x, int(size[0]), int(size[1]))


# ------------------------------------------------------------------ model graph
def model(x, n_joints):
    """main.py:29-74.  x [B,480,720,3] -> logits [B,60,90,n_joints] (one fused C call)."""
    if n_joints != engine().n_joints:
        raise ValueError('engine was configured for %d joints' % engine().n_joints)
    return engine().model(x)


def model_layerwise(x, n_joints):
    """The same graph composed op by op through the per-layer entry points, line for line
    with main.py:38-74; used by the tests to cross-check the fused `model`."""
    n_filters = np.array([64, 128, 256, 512, 512])
    if hps.debug:
        n_filters = n_filters // 4
    n_filters = [int(f) for f in n_filters]

    x1 = x
    x1 = conv_layer(x1, 5, 2, 3, n_filters[0], 'conv1_fullres')
    x1 = max_pool_layer(x1, 2, 2)
    x1 = conv_layer(x1, 5, 1, n_filters[0], n_filters[1], 'conv2_fullres')
    x1 = max_pool_layer(x1, 2, 2)
    x1 = conv_layer(x1, 5, 1, n_filters[1], n_filters[2], 'conv3_fullres')
    x1 = conv_layer(x1, 9, 1, n_filters[2], n_filters[3], 'conv4_fullres')

    x2 = resize_images(x, [int(x.shape[1]) // 2, int(x.shape[2]) // 2])
    x2 = conv_layer(x2, 5, 2, 3, n_filters[0], 'conv1_halfres')
    x2 = max_pool_layer(x2, 2, 2)
    x2 = conv_layer(x2, 5, 1, n_filters[0], n_filters[1], 'conv2_halfres')
    x2 = max_pool_layer(x2, 2, 2)
    x2 = conv_layer(x2, 5, 1, n_filters[1], n_filters[2], 'conv3_halfres')
    x2 = conv_layer(x2, 9, 1, n_filters[2], n_filters[3], 'conv4_halfres')
    x2 = resize_images(x2, [int(x1.shape[1]), int(x1.shape[2])])

    x3 = resize_images(x, [int(x.shape[1]) // 4, int(x.shape[2]) // 4])
    x3 = conv_layer(x3, 5, 2, 3, n_filters[0], 'conv1_quarterres')
    x3 = max_pool_layer(x3, 2, 2)
    x3 = conv_layer(x3, 5, 1, n_filters[0], n_filters[1], 'conv2_quarterres')
    x3 = max_pool_layer(x3, 2, 2)
    x3 = conv_layer(x3, 5, 1, n_filters[1], n_filters[2], 'conv3_quarterres')
    x3 = conv_layer(x3, 9, 1, n_filters[2], n_filters[3], 'conv4_quarterres')
    x3 = resize_images(x3, [int(x1.shape[1]), int(x1.shape[2])])

    x = x1 + x2 + x3
    x = x / 3
    x = conv_layer(x, 9, 1, n_filters[3], n_filters[4], 'conv5')
    x = conv_layer(x, 9, 1, n_filters[4], n_joints, 'conv6', last_layer=True)
    return x


def conv_mrf(A, B):
    """main.py:77-91.  A [1,120,180,1] prior, B [b,60,90,1] likelihood -> [b,60,90,1]."""
    return engine().conv_mrf(A, B)


def spatial_model(heat_map):
    """main.py:94-125.  heat_map [B,60,90,10] -> [B,60,90,9]."""
    return engine().spatial_model(heat_map)


def spatial_softmax(hm):
    """main.py:212-217."""
    return engine().spatial_softmax(hm)


def get_joints_coords(hm):
    """evaluation.py:15-24 / main.py:389-397: int32 [B,2,K] (row, col)."""
    return engine().argmax_coords(hm)


def tower(x, hm_target, use_sm=None):
    """One tower of main.py:522-531: returns (hm_pred_pd, hm_pred_sm)."""
    use_sm = hps.use_sm if use_sm is None else use_sm
    hm_pred_pd_logit = model(x, n_joints)
    hm_pred_pd = spatial_softmax(hm_pred_pd_logit)
    if use_sm:
        hm_pred_pd_with_torso = torch.cat([hm_pred_pd, hm_target[:, :, :, n_joints:]], dim=3).contiguous()
        hm_pred_sm_logit = spatial_model(hm_pred_pd_with_torso)
        hm_pred_sm = spatial_softmax(hm_pred_sm_logit)
    else:
        hm_pred_sm = hm_pred_pd                      # main.py:535
    return hm_pred_pd, hm_pred_sm


def get_different_scales(x, pad_array, crop_array, orig_h, orig_w):
    """main.py:326-348: the 4 padded + 4 cropped copies of an image, resized back to orig_h x orig_w."""
    from . import multiscale
    return multiscale.get_different_scales(engine(), x, pad_array, crop_array, orig_h, orig_w)


def scale_hm_back(hms, pad_array, crop_array, orig_h, orig_w):
    """main.py:351-379: undo the pad / crop on the 8 heat maps of an image."""
    from . import multiscale
    return multiscale.scale_hm_back(engine(), hms, pad_array, crop_array, orig_h, orig_w)


def get_predictions(X_np, Y_np, sess=None):
    """main.py:382-425 (multi-scale test-time evaluation) -> (pred_coords_pd, pred_coords_sm), each
    [2, n_joints, N]; `sess` is accepted for signature compatibility and ignored."""
    from . import multiscale
    return multiscale.get_predictions(engine(), X_np, Y_np, use_sm=hps.use_sm)


# ------------------------------------------------------------------ data set, session files (main.py:286-299,604-612,666)
DATASET_FILES = ('x_train_flic.npy', 'y_train_flic.npy', 'x_test_flic.npy', 'y_test_flic.npy')     # main.py:290-293, written by data.py
model_path = 'models_ex'                                                                            # main.py:444


def get_dataset(data_dir='.'):
    """main.py:286-294: the four arrays data.py prepares (x [N,480,720,3] fp32 in [0,1], y [N,60,90,10])."""
    missing = [f for f in DATASET_FILES if not os.path.exists(os.path.join(data_dir, f))]
    if missing:
        raise FileNotFoundError('%s not found in %r: run `python -m joint_cnn_mrf_amd.data` on the FLIC frames first (the reference\'s '
                                'data.py step), or pass --synthetic for generated data' % (', '.join(missing), data_dir))
    return tuple(np.load(os.path.join(data_dir, f), mmap_mode='r') for f in DATASET_FILES)


def get_pairwise_distr(data_dir='.'):
    """main.py:297-299: the pickle prepare_pairwise_distribution.py writes ({'<j>_<c>': [120,180] float64})."""
    import pickle
    with open(os.path.join(data_dir, 'pairwise_distribution.pickle'), 'rb') as handle:
        return pickle.load(handle)


def _synthetic_dataset(n_train, n_test):
    return (synth.make_images(n_train, seed=100), synth.make_targets(n_train, seed=200),
            synth.make_images(n_test, seed=300), synth.make_targets(n_test, seed=400))


def initial_params(args, pairwise_distr):
    """tf.global_variables_initializer on the graph of main.py:474-487: He-initialised convolutions, identity BatchNorm,
    energies = the pairwise distributions, biases = 1e-5."""
    params = synth.make_pd_params(debug=args.debug)
    if args.use_sm:
        params.update(synth.make_sm_params(pairwise_distr, kind='init'))
    return params


def restore_params(path, args):
    """saver.restore (main.py:612): `path` is a tf.train.Saver checkpoint prefix (P.index + P.data-00000-of-00001, read by
    tf_checkpoint.py) or an .npz keyed by the same variable names.  Returns every saved variable, optimizer slots included."""
    from . import checkpoint, tf_checkpoint
    if path.endswith('.npz'):
        with np.load(path) as z:
            state = {k: z[k] for k in z.files}
    else:
        state = tf_checkpoint.load_checkpoint(path)
    checkpoint.validate({k: v for k, v in state.items() if k in checkpoint.expected_shapes(args.debug, args.use_sm)}, args.debug, args.use_sm)
    return state


class TowerTrainer:
    """The training side of the in-process towers (main.py:509-577): every listed device runs compute_gradients on its
    slice of the batch, the tower gradients are averaged (average_gradients, main.py:243-267: here a sum on the first
    device divided by the tower count) and the same clipped update is applied to every replica.  The BatchNorm moving
    statistics are shared variables in the reference and the towers' update ops (main.py:557) run on them one after the
    other, tower 0 first: moving <- 0.9 * moving + 0.1 * stat_i for i = 0, 1, ...  Each replica here did that once from the
    common start, so the composition is rebuilt from the per-replica results and written to every replica."""
    BN_DECAY = 0.9          # decay=0.9 of the reference's tf.contrib.layers.batch_norm calls (main.py:113,129)

    def __init__(self, towers, params, **trainer_kw):
        from .train import Trainer
        self.towers = towers
        self.trainers = [Trainer(e, **trainer_kw) for e in towers.engines]
        self.moving = {k: np.asarray(v, np.float32).reshape(-1).copy() for k, v in params.items()
                       if k.endswith('moving_mean') or k.endswith('moving_variance')}

    def train_step(self, x, y):
        tw = self.towers
        for tr, eng, (lo, hi) in zip(self.trainers, tw.engines, tw.slices(x.shape[0])):
            xs = torch.as_tensor(x[lo:hi]).to(eng.device, non_blocking=True).contiguous()
            ys = torch.as_tensor(y[lo:hi]).to(eng.device, non_blocking=True).contiguous()
            with torch.cuda.device(eng.device):
                tr.loss_and_grads(xs, ys)
        dev0 = tw.engines[0].device
        if tw.n > 1:
            total = self.trainers[0].grads
            for tr in self.trainers[1:]:
                total += tr.grads.to(dev0)
            total /= tw.n
            for tr in self.trainers[1:]:
                tr.grads.copy_(total)
        for tr, eng in zip(self.trainers, tw.engines):
            with torch.cuda.device(eng.device):
                tr.apply()
        if tw.n == 1:                                              # one tower: its own moving statistics are the result, nothing to compose or read back
            return self.trainers[0].losses
        names = sorted(self.moving)
        for i, name in enumerate(names):
            start = self.moving[name]
            cur = start.astype(np.float64)
            for tr in self.trainers:                               # r_i = d * start + (1 - d) * stat_i  ->  cur = d * cur + (r_i - d * start)
                cur = self.BN_DECAY * cur + (tr.get_tensor(name, start.shape).astype(np.float64) - self.BN_DECAY * start)
            self.moving[name] = cur.astype(np.float32)
            for eng in tw.engines:
                eng.update_tensor(name, self.moving[name], refresh=i == len(names) - 1)
        return self.trainers[0].losses


def train_main(args):
    """`--train` (main.py:620-667): the reference's epoch loop -- shuffled whole batches (get_next_batch), eval_error on
    the first n_eval_ex train / test examples after every epoch, the reference's progress line, a checkpoint per epoch once
    half of the epochs are done (tf.train.Saver format) -- on one tower per device of --gpus."""
    from . import checkpoint, evaluation, tf_checkpoint
    from .dist import Towers
    from .train import Trainer
    if args.data_augm:
        raise NotImplementedError('--data_augm: the augmentation pipeline (augmentation.py: flips, rotations, crops on the TF input '
                                  'queue) is outside the hot path this build covers; train without it')
    t_start = time.time()
    if args.synthetic:
        x_train, y_train, x_test, y_test = _synthetic_dataset(args.synthetic_size, max(args.batch_size, args.synthetic_size // 2))
        pairwise = synth.synthetic_priors()
    else:
        x_train, y_train, x_test, y_test = get_dataset(args.data_dir)
        pairwise = get_pairwise_distr(args.data_dir)
    n_train, n_test = x_train.shape[0], x_test.shape[0]
    rng = np.random.RandomState(args.seed)
    n_eval_ex = 512 if args.debug else 1100                              # main.py:453
    if args.debug and not args.synthetic:                                # main.py:459-462
        n_train, n_test = min(1024, n_train), min(512, n_test)
        tr_idx, te_idx = np.sort(rng.permutation(x_train.shape[0])[:n_train]), np.sort(rng.permutation(x_test.shape[0])[:n_test])
        x_train, y_train, x_test, y_test = x_train[tr_idx], y_train[tr_idx], x_test[te_idx], y_test[te_idx]
    n_updates_total = args.n_epochs * n_train // args.batch_size        # main.py:466
    state = restore_params(args.restore_path, args) if args.restore else None
    params = {k: v for k, v in state.items() if k in checkpoint.expected_shapes(args.debug, args.use_sm)} if state else initial_params(args, pairwise)
    towers = Towers(params, args.gpus, precision=args.precision)
    tt = TowerTrainer(towers, params, optimizer=args.optimizer, lr=args.lr, lmbd=args.lmbd, use_sm=args.use_sm, n_updates_total=n_updates_total)
    if state:
        for tr in tt.trainers:
            checkpoint.restore_session_state(tr, state)
    eng = towers.engines[0]
    model_name = '{}_lr={}_lambda={}_bs={}'.format(time.strftime('%Y-%m-%d %H:%M:%S'), args.lr, args.lmbd, args.batch_size)     # main.py:447
    joints_to_eval, det_radius = [2], 10                                 # main.py:455-456

    def report(epoch):
        tr_e = evaluation.eval_error(x_train[:n_eval_ex], y_train[:n_eval_ex], eng, args.batch_size, args.use_sm, joints_to_eval, det_radius)
        te_e = evaluation.eval_error(x_test[:n_eval_ex], y_test[:n_eval_ex], eng, args.batch_size, args.use_sm, joints_to_eval, det_radius)
        print('Epoch {:d}  test_dr {:.3f} {:.3f}  train_dr {:.3f} {:.3f}  test_mse {:.5f} {:.5f}  train_mse {:.5f} {:.5f}'.format(
            epoch, te_e[2], te_e[3], tr_e[2], tr_e[3], te_e[0], te_e[1], tr_e[0], tr_e[1]), flush=True)      # main.py:628-631,656-657

    report(0)
    for epoch in range(1, args.n_epochs + 1):
        for bx, by in evaluation.get_next_batch(x_train, y_train, args.batch_size, shuffle=True, rng=rng):      # main.py:641
            tt.train_step(np.ascontiguousarray(bx, np.float32), np.ascontiguousarray(by, np.float32))
        report(epoch)
        if epoch > args.n_epochs // 2:                                   # main.py:663-666
            tf_checkpoint.save_checkpoint('{}/{}-{}'.format(args.model_path, model_name, epoch), checkpoint.session_state(tt.trainers[0], params))
    print('Done in {:.2f} min\n\n'.format((time.time() - t_start) / 60))
    return tt


def main(argv=None):
    global hps
    args = build_parser().parse_args(argv)
    hps = args
    if args.restore and not args.restore_path:
        raise SystemExit('--restore needs --restore_path <checkpoint prefix or .npz> (the reference hard-codes best_model_name, main.py:443)')
    for g in args.gpus:
        if g < 0 or g >= torch.cuda.device_count():
            raise SystemExit('--gpus %s: device %d does not exist (%d visible)' % (args.gpus, g, torch.cuda.device_count()))
    if args.train:
        train_main(args)
        return
    # evaluation run (main.py:668-675): multi-scale predictions of the test set -> matlab/predictions.mat
    from . import checkpoint
    from .dist import Towers
    if args.synthetic:
        _xt, _yt, x_test, y_test = _synthetic_dataset(args.batch_size, args.synthetic_size)
        pairwise = synth.synthetic_priors()
    else:
        _xt, _yt, x_test, y_test = get_dataset(args.data_dir)
        pairwise = get_pairwise_distr(args.data_dir)
    state = restore_params(args.restore_path, args) if args.restore else None
    params = {k: v for k, v in state.items() if k in checkpoint.expected_shapes(args.debug, args.use_sm)} if state else initial_params(args, pairwise)
    # one set of engines only (each fp32 engine caches multi-GB filter spectra): the module engine for the multi-scale wrapper, which runs
    # on one device, or one tower per listed device for the single-scale run
    towers = None
    t0 = time.time()
    if args.multiscale:
        if len(args.gpus) > 1:
            print('--multiscale evaluates on device %d only; the other --gpus entries are not used' % args.gpus[0], file=sys.stderr)
        configure(params, device=args.gpus[0], precision=args.precision, debug=args.debug)
        pred_pd, pred_sm = get_predictions(np.asarray(x_test), np.asarray(y_test))                     # main.py:674
    else:                                                                                              # single scale, sharded over the towers
        hps.debug = bool(args.debug)
        towers = Towers(params, args.gpus, precision=args.precision)
        B = args.batch_size
        pd, sm = [], []
        for lo in range(0, (x_test.shape[0] // B) * B, B):
            r = towers.forward(np.ascontiguousarray(x_test[lo:lo + B], np.float32), np.ascontiguousarray(y_test[lo:lo + B, :, :, n_joints:], np.float32),
                               use_sm=args.use_sm)
            pd.append(r['pd_coords'])
            sm.append(r['sm_coords'] if args.use_sm else r['pd_coords'])
        to_ref = lambda c: torch.cat(c).permute(1, 2, 0).cpu().numpy()      # [2,K,N] (row, col) stacked on the last axis, main.py:425
        pred_pd, pred_sm = to_ref(pd), to_ref(sm)
    torch.cuda.synchronize()
    dt = time.time() - t0
    if args.predictions:
        import scipy.io
        os.makedirs(os.path.dirname(args.predictions) or '.', exist_ok=True)
        scipy.io.savemat(args.predictions, {'flic_pred_pd': pred_pd, 'flic_pred_sm': pred_sm})         # main.py:675
    print(json.dumps({'n_images': int(pred_pd.shape[2]), 'gpus': args.gpus, 'use_sm': bool(args.use_sm), 'debug': bool(args.debug),
                      'multiscale': bool(args.multiscale), 'seconds': dt, 'images_per_sec': pred_pd.shape[2] / dt,
                      'coords_image0_pd': pred_pd[:, :, 0].tolist()}))
    if towers is not None:
        towers.close()


if __name__ == '__main__':
    main()
