"""CPU restatement of the reference's joint training step.

TEST INFRASTRUCTURE (see oracle/__init__.py) -- PARITY UNPINNED: TensorFlow is absent, so the
TF-1.x semantics below are restated from its documented behaviour, not checked against it.

What one `sess.run(train_step, {flag_train: True})` does (main.py:511-577,644):
  * forward in training mode: every BatchNorm normalises with the *batch* mean / biased variance
    (tf.contrib.layers.batch_norm, fused, eps 1e-3; main.py:113,129) and its update op moves
    moving_mean / moving_variance with decay 0.9, the variance with Bessel's correction
    (fused_batch_norm's running estimate);
  * loss = CE(pd_logits, target) + CE(sm_logits, target) + lmbd * sum_{'weights'} sum(w^2)/2
    (main.py:220-240,195-205,538-541); CE = mean over (image, joint) of the soft-label
    cross-entropy with a softmax over the 5400 pixels;
  * gradients of that loss w.r.t. every trainable variable (conv weights/biases, BN gamma/beta,
    bn_sm gamma/beta, the 81 energies and 81 biases); the spatial-model loss also flows back
    into the part detector through hm_pred_pd (main.py:523,528 -- no stop_gradient);
  * tower average (main.py:243-267), clip_by_global_norm 4.0 (main.py:302-309,576), then
    tf.train.AdamOptimizer / MomentumOptimizer (main.py:501-506,577).

The graph is written with differentiable torch-CPU ops in float64 and differentiated by
autograd; max-pool ties route the gradient to the first maximum in window order, as TF's
MaxPoolGrad and torch's both do.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from .jcm_oracle import JOINT_NAMES, JOINT_DEPENDENCE, N_JOINTS, BN_EPS, SM_DELTA
from .jcm_oracle_torch import conv2d_same, max_pool_same, resize_bilinear_tf1, softplus5

BN_DECAY = 0.9            # main.py:113,129
CLIP_NORM = 4.0           # main.py:576
ADAM_B1, ADAM_B2, ADAM_EPS = 0.9, 0.999, 1e-8   # tf.train.AdamOptimizer defaults
MOMENTUM = 0.9            # main.py:504


def is_trainable(name):
    return not (name.endswith('moving_mean') or name.endswith('moving_variance'))


def to_torch(params, dtype=torch.float64):
    """numpy dict -> torch dict; trainable entries become autograd leaves."""
    out = {}
    for k, v in params.items():
        t = torch.as_tensor(np.ascontiguousarray(v)).to(dtype).clone()
        if is_trainable(k):
            t.requires_grad_(True)
        out[k] = t
    return out


def _bn_train(x, p, scope, stats):
    """Fused batch norm in training mode on an NCHW tensor; records the batch statistics."""
    g, b = p[scope + '/BatchNorm/gamma'], p[scope + '/BatchNorm/beta']
    n = x.shape[0] * x.shape[2] * x.shape[3]
    mean = x.mean(dim=(0, 2, 3))
    var = ((x - mean.view(1, -1, 1, 1)) ** 2).mean(dim=(0, 2, 3))              # biased: used to normalise
    stats[scope] = (mean.detach(), (var * (n / max(n - 1, 1))).detach())       # unbiased: feeds the moving average
    sh = (1, -1, 1, 1)
    return (x - mean.view(sh)) * (g * torch.rsqrt(var + BN_EPS)).view(sh) + b.view(sh)


def _conv_layer(x, p, stride, name, stats, last_layer=False):
    z = conv2d_same(x, p[name + '/weights'], stride) + p[name + '/biases'].view(1, -1, 1, 1)
    return z if last_layer else _bn_train(F.relu(z), p, name, stats)


def model_train(x_nchw, p, stats):
    """main.py:29-74 with flag_train=True."""
    H, W = x_nchw.shape[2], x_nchw.shape[3]

    def branch(h, res):
        h = max_pool_same(_conv_layer(h, p, 2, 'conv1_' + res, stats))
        h = max_pool_same(_conv_layer(h, p, 1, 'conv2_' + res, stats))
        h = _conv_layer(h, p, 1, 'conv3_' + res, stats)
        return _conv_layer(h, p, 1, 'conv4_' + res, stats)

    x1 = branch(x_nchw, 'fullres')
    x2 = resize_bilinear_tf1(branch(resize_bilinear_tf1(x_nchw, H // 2, W // 2), 'halfres'), x1.shape[2], x1.shape[3])
    x3 = resize_bilinear_tf1(branch(resize_bilinear_tf1(x_nchw, H // 4, W // 4), 'quarterres'), x1.shape[2], x1.shape[3])
    h = (x1 + x2 + x3) / 3
    h = _conv_layer(h, p, 1, 'conv5', stats)
    return _conv_layer(h, p, 1, 'conv6', stats, last_layer=True)          # NCHW logits


def conv_mrf_t(prior_hw, lik_bhw):
    """main.py:77-91, differentiable: the VALID true convolution is a cross-correlation of the
    prior with the flipped likelihood maps as B output channels, then the 61x91 -> 60x90 resize."""
    k = torch.flip(lik_bhw, dims=(1, 2))[:, None]                        # [B,1,60,90]
    pre = F.conv2d(prior_hw[None, None], k)                              # [1,B,61,91]
    return resize_bilinear_tf1(pre, lik_bhw.shape[1], lik_bhw.shape[2])[0]   # [B,60,90]


def spatial_model_train(hm10_nchw, p, stats, n_joints=N_JOINTS):
    """main.py:94-125 with flag_train=True -> [B,K,60,90] logits."""
    hm = _bn_train(hm10_nchw, p, 'bn_sm', stats)
    out = []
    for jid, jname in enumerate(JOINT_NAMES[:n_joints]):
        e = torch.log(softplus5(hm[:, jid]) + SM_DELTA)
        for cname in JOINT_DEPENDENCE[jname]:
            cid = JOINT_NAMES.index(cname)
            prior = softplus5(p['energy_%s_%s' % (jname, cname)])[0, :, :, 0]
            bias = softplus5(p['bias_%s_%s' % (jname, cname)])[0, :, :, 0]
            e = e + torch.log(conv_mrf_t(prior, softplus5(hm[:, cid])) + bias + SM_DELTA)
        out.append(e)
    return torch.stack(out, dim=1)


class _TfSoftmaxCE(torch.autograd.Function):
    """tf.nn.softmax_cross_entropy_with_logits as TF-1.x computes it: loss = -sum(labels * log_softmax(logits)); the
    gradient w.r.t. the logits that the op's kernel emits ("backprop") is softmax - labels, multiplied by the upstream
    gradient (tensorflow/python/ops/nn_grad.py: _SoftmaxCrossEntropyWithLogitsGrad).  That is the exact derivative only
    when the labels of a row sum to one; FLIC target blobs clipped by the map border (data.py:171-183) do not, and the
    reference trains with TF's gradient -- so the restatement must, too.  No gradient flows to the labels (constants)."""

    @staticmethod
    def forward(ctx, logits, labels):            # [..., n] over the last axis
        ls = torch.log_softmax(logits, dim=-1)
        ctx.save_for_backward(ls, labels)
        return -(labels * ls).sum(dim=-1)

    @staticmethod
    def backward(ctx, g):
        ls, labels = ctx.saved_tensors
        return g.unsqueeze(-1) * (ls.exp() - labels), None


def softmax_cross_entropy(logits_nchw, target_nchw):
    """main.py:220-240: softmax over the pixels, soft labels, mean over (image, joint); TF's gradient (see _TfSoftmaxCE)."""
    B, K = logits_nchw.shape[:2]
    return _TfSoftmaxCE.apply(logits_nchw.reshape(B, K, -1), target_nchw.reshape(B, K, -1)).mean()


def weight_decay(p):
    """main.py:195-205 with var_pattern='weights': sum of tf.nn.l2_loss = sum(w^2)/2."""
    return sum((v ** 2).sum() / 2 for k, v in p.items() if 'weights' in k)


def loss_and_grads(x_nhwc, y_nhwc, params, use_sm=True, lmbd=0.001, n_joints=N_JOINTS, dtype=torch.float64):
    """One tower's loss_tower and compute_gradients (main.py:538-541,559-560).

    Returns dict: loss, loss_pd, loss_sm, l2, grads{name: ndarray}, bn_stats{scope: (mean, unbiased var)},
    pd_logits / sm_logits (NHWC ndarrays)."""
    p = to_torch(params, dtype)
    x = torch.as_tensor(np.ascontiguousarray(x_nhwc)).to(dtype).permute(0, 3, 1, 2).contiguous()
    y = torch.as_tensor(np.ascontiguousarray(y_nhwc)).to(dtype).permute(0, 3, 1, 2).contiguous()
    stats = {}
    pd_logits = model_train(x, p, stats)
    B, K = pd_logits.shape[:2]
    loss_pd = softmax_cross_entropy(pd_logits, y[:, :n_joints])
    r = {'pd_logits': pd_logits.detach().permute(0, 2, 3, 1).numpy()}
    if use_sm:
        pd_prob = torch.softmax(pd_logits.reshape(B, K, -1), dim=2).reshape(pd_logits.shape)
        hm10 = torch.cat([pd_prob, y[:, n_joints:]], dim=1)                  # main.py:528
        sm_logits = spatial_model_train(hm10, p, stats, n_joints)
        loss_sm = softmax_cross_entropy(sm_logits, y[:, :n_joints])
        r['sm_logits'] = sm_logits.detach().permute(0, 2, 3, 1).numpy()
    else:
        loss_sm = softmax_cross_entropy(pd_logits, y[:, :n_joints])         # main.py:535: same logits
    l2 = weight_decay(p)
    loss = loss_pd + loss_sm + lmbd * l2
    names = [k for k in sorted(p) if p[k].requires_grad]
    gs = torch.autograd.grad(loss, [p[k] for k in names], allow_unused=True)
    r.update(loss=float(loss.detach()), loss_pd=float(loss_pd.detach()), loss_sm=float(loss_sm.detach()), l2=float(l2.detach()),
             grads={k: (np.zeros(tuple(p[k].shape)) if g is None else g.numpy()) for k, g in zip(names, gs)},
             bn_stats={k: (m.numpy(), v.numpy()) for k, (m, v) in stats.items()})
    return r


def update_moving(params, bn_stats, decay=BN_DECAY):
    """The UPDATE_OPS of main.py:557: moving = decay*moving + (1-decay)*batch."""
    out = {}
    for scope, (m, v) in bn_stats.items():
        for name, val in (('moving_mean', m), ('moving_variance', v)):
            k = '%s/BatchNorm/%s' % (scope, name)
            out[k] = np.asarray(params[k], np.float64) * decay + (1 - decay) * val
    return out


def global_norm(grads):
    return math.sqrt(sum(float((np.asarray(g, np.float64) ** 2).sum()) for g in grads.values()))


def clip_by_global_norm(grads, clip=CLIP_NORM):
    """tf.clip_by_global_norm (main.py:302-309): g * clip / max(norm, clip)."""
    n = global_norm(grads)
    s = clip / max(n, clip)
    return {k: np.asarray(g, np.float64) * s for k, g in grads.items()}, n


def adam_apply(params, grads, slots, step, lr):
    """tf.train.AdamOptimizer.apply_gradients for update number `step` (1-based):
    lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m,v moments; var -= lr_t*m/(sqrt(v)+eps)."""
    lr_t = lr * math.sqrt(1 - ADAM_B2 ** step) / (1 - ADAM_B1 ** step)
    new = {}
    for k, g in grads.items():
        m = slots.setdefault(k + '/Adam', np.zeros_like(g, np.float64))
        v = slots.setdefault(k + '/Adam_1', np.zeros_like(g, np.float64))
        m[...] = ADAM_B1 * m + (1 - ADAM_B1) * g
        v[...] = ADAM_B2 * v + (1 - ADAM_B2) * g * g
        new[k] = np.asarray(params[k], np.float64) - lr_t * m / (np.sqrt(v) + ADAM_EPS)
    return new


def momentum_apply(params, grads, slots, lr):
    """tf.train.MomentumOptimizer(momentum=0.9): acc = 0.9*acc + g; var -= lr*acc."""
    new = {}
    for k, g in grads.items():
        a = slots.setdefault(k + '/Momentum', np.zeros_like(g, np.float64))
        a[...] = MOMENTUM * a + g
        new[k] = np.asarray(params[k], np.float64) - lr * a
    return new


def piecewise_lr(n_iters, n_updates_total, lr):
    """main.py:467-469,492: tf.train.piecewise_constant(n_iters, [.7,.8,.9]*total, [lr, lr/2, lr/5, lr/10])."""
    bounds = [round(0.7 * n_updates_total), round(0.8 * n_updates_total), round(0.9 * n_updates_total)]
    vals = [lr, lr / 2, lr / 5, lr / 10]
    for b, v in zip(bounds, vals):
        if n_iters <= b:
            return v
    return vals[-1]


def train_step(x, y, params, slots, step, lr=0.001, lmbd=0.001, use_sm=True, optimizer='adam', towers=1):
    """One full update on `towers` equal batch slices (main.py:511-577).  Mutates nothing;
    returns (new_params, info).  BN moving statistics are updated tower after tower, as the
    per-tower update ops of main.py:557 do (order of towers = GPU index)."""
    B = x.shape[0] // towers
    acc, infos = None, []
    cur = dict(params)
    for t in range(towers):
        r = loss_and_grads(x[t * B:(t + 1) * B], y[t * B:(t + 1) * B], params, use_sm=use_sm, lmbd=lmbd)
        cur.update(update_moving(cur, r['bn_stats']))
        acc = r['grads'] if acc is None else {k: acc[k] + r['grads'][k] for k in acc}
        infos.append(r)
    grads = {k: g / towers for k, g in acc.items()}                      # average_gradients
    clipped, norm = clip_by_global_norm(grads)
    if optimizer == 'adam':
        upd = adam_apply(params, clipped, slots, step, lr)
    elif optimizer == 'momentum':
        upd = momentum_apply(params, clipped, slots, lr)
    else:
        raise Exception('wrong optimizer')                               # main.py:506
    cur.update(upd)
    return cur, {'loss': float(np.mean([i['loss'] for i in infos])), 'grad_norm': norm, 'towers': infos, 'grads': grads}
