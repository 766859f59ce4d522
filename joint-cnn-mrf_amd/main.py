"""Host-side mirror of the reference's main.py for the joint-heat-map inference path.

Same function names, argument order and NHWC layouts as /root/reference/main.py, so a
caller of `model`, `conv_mrf`, `spatial_model`, `spatial_softmax` (graph at main.py:522-531)
switches by importing this module; tensors are torch-ROCm instead of tf.Tensor, and the
arithmetic runs in libjcm's HIP kernels (no TensorFlow, no CPU fallback).

Where the reference reads module globals (`hps`, `flag_train`, `pairwise_energies`,
`pairwise_biases`, `joint_names`, `joint_dependence`, `n_joints`, `hm_height`, `hm_width`)
this module keeps the same names; they are filled by `configure()` instead of at import
(the reference parses argv and loads `.npy` files at import time, main.py:440,459).

CLI: `python -m joint_cnn_mrf_amd.main --gpus 0 --use_sm --batch_size 64 [--debug]` runs the
inference tower on synthetic data; `--train` runs the reference's epoch loop (main.py:620-667) on a
synthetic data set through `train.Trainer`.
"""
import argparse
import json
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
    """The reference's flags (main.py:428-439); training-only ones are accepted and ignored."""
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


def train_main(args):
    """`--train` (main.py:620-667): n_epochs over a synthetic data set of 4 batches; prints the reference's
    per-epoch line.  Losses are the training-mode losses of the last batch of the epoch; detection rates
    are evaluated in inference mode (flag_train=False) on the first batch."""
    from . import evaluation
    from .train import Trainer
    dev_id = args.gpus[0]
    torch.cuda.set_device(dev_id)
    params = synth.make_pd_params(debug=args.debug)
    if args.use_sm:
        params.update(synth.make_sm_params(synth.synthetic_priors(), kind='init'))
    eng = configure(params, device=dev_id, debug=args.debug)
    B = args.batch_size
    n_train = 4 * B
    n_updates_total = args.n_epochs * n_train // B                      # main.py:466
    tr = Trainer(eng, optimizer=args.optimizer, lr=args.lr, lmbd=args.lmbd, use_sm=args.use_sm, n_updates_total=n_updates_total)
    X = [torch.as_tensor(synth.make_images(B, seed=100 + i), device=eng.device) for i in range(n_train // B)]
    Y = [torch.as_tensor(synth.make_targets(B, seed=200 + i), device=eng.device) for i in range(n_train // B)]
    joints_to_eval, det_radius = [2], 10                                # main.py:455-456
    for epoch in range(1, args.n_epochs + 1):
        t0 = time.time()
        for xb, yb in zip(X, Y):
            losses, _ = tr.train_step(xb, yb)
        l = losses.cpu().numpy()
        r = eng.forward(X[0], Y[0][..., n_joints:].contiguous(), use_sm=args.use_sm)
        tgt = Y[0][..., :n_joints].contiguous()
        dr_pd = evaluation.det_rate(r['pd_prob'], tgt, det_radius, joints_to_eval, engine=eng)
        dr_sm = evaluation.det_rate(r['sm_prob'] if args.use_sm else r['pd_prob'], tgt, det_radius, joints_to_eval, engine=eng)
        print('Epoch {:d}  train_dr {:.3f} {:.3f}  train_loss {:.5f} {:.5f}  total {:.5f}  lr {:g}  {:.1f} img/s'.format(
            epoch, dr_pd, dr_sm, l[1], l[2], l[0], piecewise(tr), n_train / (time.time() - t0)))
    return tr


def piecewise(tr):
    from .train import piecewise_lr
    return piecewise_lr(tr.n_iters, tr.n_updates_total, tr.lr)


def main(argv=None):
    global hps
    args = build_parser().parse_args(argv)
    if args.train:
        hps = args
        train_main(args)
        return
    if args.restore:
        raise SystemExit('--restore: no TF checkpoint ships with the reference; parameters are synthetic here')
    hps = args
    dev = args.gpus[0]
    torch.cuda.set_device(dev)
    params = synth.make_pd_params(debug=args.debug)
    if args.use_sm:
        params.update(synth.make_sm_params(synth.synthetic_priors(), kind='init'))
    eng = configure(params, device=dev, debug=args.debug)
    B = args.batch_size
    x = torch.as_tensor(synth.make_images(B), device=eng.device)
    torso = torch.as_tensor(synth.make_torso(B), device=eng.device)
    eng.forward(x, torso, use_sm=args.use_sm, want_prob=False)
    torch.cuda.synchronize()
    t0 = time.time()
    r = eng.forward(x, torso, use_sm=args.use_sm, want_prob=False)
    torch.cuda.synchronize()
    dt = time.time() - t0
    key = 'sm_coords' if args.use_sm else 'pd_coords'
    print(json.dumps({'batch_size': B, 'use_sm': bool(args.use_sm), 'debug': bool(args.debug), 'seconds': dt,
                      'images_per_sec': B / dt, 'coords_image0': r[key][0].cpu().numpy().tolist()}))


if __name__ == '__main__':
    main()
