"""Host side of the joint training step (reference: main.py:464-470,491-506,511-577,644).

`Trainer` wraps an fp32 `Engine`: `loss_and_grads` is one tower's forward (training-mode BatchNorm)
+ `opt.compute_gradients`; `train_step` adds the tower average (an RCCL all-reduce of the flat
gradient buffer when `torch.distributed` is initialised -- average_gradients, main.py:243-267),
`grad_renorm(., 4.0)` and `opt.apply_gradients`.  The arithmetic is libjcm's HIP kernels; torch
only owns the buffers and runs the collective.
"""
import ctypes

import numpy as np
import torch

from . import _lib

CLIP_NORM = 4.0      # main.py:576


def piecewise_lr(n_iters, n_updates_total, lr):
    """lr_tf of main.py:467-469,492: tf.train.piecewise_constant(n_iters, round([.7,.8,.9]*total),
    [lr, lr/2, lr/5, lr/10]) -- the value used by update number n_iters+1."""
    bounds = [round(0.7 * n_updates_total), round(0.8 * n_updates_total), round(0.9 * n_updates_total)]
    vals = [lr, lr / 2, lr / 5, lr / 10]
    for b, v in zip(bounds, vals):
        if n_iters <= b:
            return v
    return vals[-1]


class Trainer:
    def __init__(self, engine, optimizer='adam', lr=0.001, lmbd=0.001, use_sm=True, n_updates_total=None, overlap_allreduce=False):
        # fp32 handle: fp32 tensors (frequency domain by default; the fp32 MFMA chain with conv9_fft=False; fp16x3 direct kernels with f32_conv='split16'); bf16 handle: mixed precision
        # (bf16 activations / gradients and bf16 MFMA, fp32 master weights, statistics, losses, spatial model, optimizer)
        if optimizer not in ('adam', 'momentum'):
            raise Exception('wrong optimizer')                      # main.py:506
        self.eng = engine
        self.optimizer, self.lr, self.lmbd, self.use_sm = optimizer, float(lr), float(lmbd), bool(use_sm)
        self.n_updates_total = n_updates_total
        # start each layer's gradient all-reduce (RCCL) as soon as the backward pass has produced it; opt-in.  The boxes
        # of this build have one GPU: the path runs end to end on a one-rank RCCL group (tests/test_gpu_dist.py)
        self.overlap_allreduce = bool(overlap_allreduce)
        self._cb_error = None
        self._lib = engine._lib
        _lib.check(self._lib.jcm_train_begin(engine._h), 'jcm_train_begin')
        nt, ne = ctypes.c_int64(), ctypes.c_int64()
        _lib.check(self._lib.jcm_train_param_count(engine._h, ctypes.byref(nt), ctypes.byref(ne)), 'jcm_train_param_count')
        self.n_elements = ne.value
        self.layout = []                                            # (name, offset, count)
        buf = ctypes.create_string_buffer(256)
        for i in range(nt.value):
            off, cnt = ctypes.c_int64(), ctypes.c_int64()
            _lib.check(self._lib.jcm_train_param_info(engine._h, i, buf, 256, ctypes.byref(off), ctypes.byref(cnt)),
                       'jcm_train_param_info')
            self.layout.append((buf.value.decode(), off.value, cnt.value))
        self.grads = torch.zeros(self.n_elements, dtype=torch.float32, device=engine.device)
        self._moving = None                                         # [(name, offset, count)] of the BN moving statistics
        self._ready_hook = None                                     # user hook(offset, count) for the gradient-ready notifications
        self._pending, self._covered, self._side = [], [], None
        self._cb = _lib.GRAD_READY_FN(self._on_grads_ready)         # keep the ctypes thunk alive as long as the trainer
        _lib.check(self._lib.jcm_train_set_grad_callback(engine._h, ctypes.cast(self._cb, ctypes.c_void_p), None),
                   'jcm_train_set_grad_callback')
        self.losses = torch.zeros(4, dtype=torch.float32, device=engine.device)

    @property
    def n_iters(self):
        n = ctypes.c_int64()
        _lib.check(self._lib.jcm_train_steps(self.eng._h, ctypes.byref(n)), 'jcm_train_steps')
        return n.value

    def loss_and_grads(self, x, y):
        """x [B,H,W,3], y [B,60,90,K+1] -> (losses[4] device tensor, flat grads device tensor).
        losses = (loss_tower, loss_pd, loss_sm, weight_decay)."""
        e = self.eng
        e._chk(x, 4, 'x')
        e._chk(y, 4, 'y')
        B, H, W, C = x.shape
        if C != 3 or y.shape[0] != B or y.shape[3] != e.n_joints + 1:
            raise ValueError('x must be [B,H,W,3] and y [B,h,w,%d]; got %s, %s' % (e.n_joints + 1, tuple(x.shape), tuple(y.shape)))
        self._cb_error = None
        status = self._lib.jcm_train_loss_grads(e._h, e._p(x), e._p(y), B, H, W, int(self.use_sm), self.lmbd,
                                                e._p(self.grads), e._p(self.losses))
        if self._cb_error is not None:      # ctypes swallows exceptions raised inside a callback: re-raise them here
            err, self._cb_error = self._cb_error, None
            self._pending, self._covered = [], []
            raise err
        _lib.check(status, 'jcm_train_loss_grads')
        return self.losses, self.grads

    def layer_grads(self, scope, x, dz, want_dx=True):
        """The weight gradient (+ lmbd * w) and the data gradient of ONE stride-1 conv layer on given tensors, through the kernels the training
        step uses on this engine: x [B,H,W,Cin], dz [B,H,W,Cout] -> (dw flat numpy [k*k*Cin*Cout] in HWIO order, dx [B,H,W,Cin] device or None)."""
        e = self.eng
        e._chk(x, 4, 'x')
        e._chk(dz, 4, 'dz')
        B, H, W, _ = x.shape
        dx = torch.empty_like(x) if want_dx else None
        _lib.check(self._lib.jcm_train_layer_grads(e._h, scope.encode(), e._p(x), e._p(dz), B, H, W, self.lmbd, e._p(self.grads), e._p(dx)),
                   'jcm_train_layer_grads(%s)' % scope)
        off, cnt = next((o, c) for n, o, c in self.layout if n == scope + '/weights')
        return self.grads[off:off + cnt].cpu().numpy(), dx

    def grads_dict(self):
        """The flat gradient buffer as {TF variable name: numpy array} (reference shapes unknown here: flat)."""
        g = self.grads.cpu().numpy()
        return {n: g[o:o + c].copy() for n, o, c in self.layout}

    # ------------------------------------------------------------------ tower average overlapped with the backward pass
    def set_ready_hook(self, fn):
        """fn(offset, count) is called from inside loss_and_grads as each layer's gradients become final."""
        self._ready_hook = fn

    def _overlap_active(self):
        import torch.distributed as dist
        return self.overlap_allreduce and dist.is_available() and dist.is_initialized() and dist.get_backend() == 'nccl'

    def _on_grads_ready(self, _user, offset, count):
        """jcm_grad_ready_fn: the kernels writing grads[offset:offset+count] are enqueued on the engine's stream.  With RCCL
        the range's all-reduce starts now, on a side stream behind an event, while the backward pass keeps the compute
        stream busy (xGMI links and MFMA pipes are independent resources); other back ends reduce after the pass."""
        if self._cb_error is not None:
            return                                          # a previous notification failed: stop enqueueing, re-raise after the pass
        try:
            if self._ready_hook is not None:
                self._ready_hook(int(offset), int(count))
            if not self._overlap_active():
                return
            import torch.distributed as dist
            if self._side is None:
                self._side = torch.cuda.Stream(device=self.eng.device)
            ev = torch.cuda.Event()
            ev.record(self.eng._stream)
            self._side.wait_event(ev)
            with torch.cuda.stream(self._side):
                self._pending.append(dist.all_reduce(self.grads[offset:offset + count], async_op=True))
            self._covered.append((int(offset), int(count)))
        except BaseException as exc:                        # noqa: BLE001 -- must not propagate into the C caller
            self._cb_error = exc

    def average_gradients(self):
        """main.py:243-267 across ranks: mean of the per-tower gradients.  Ranges already in flight (see _on_grads_ready)
        are waited for; whatever was not reported during the pass is reduced now."""
        from . import dist as jdist
        if not self._pending:
            jdist.average_gradients(self.grads, stream=self.eng._stream)
            return
        import torch.distributed as dist
        with torch.cuda.stream(self.eng._stream):
            for w in self._pending:
                w.wait()                                    # the compute stream waits for the collective
            pos = 0
            for off, cnt in sorted(self._covered) + [(self.n_elements, 0)]:
                if off > pos:
                    dist.all_reduce(self.grads[pos:off])    # e.g. the spatial-model blocks when use_sm is off (zeros)
                pos = max(pos, off + cnt)
            if dist.get_world_size() > 1:
                self.grads.div_(dist.get_world_size())
        self._pending, self._covered = [], []

    def apply(self, lr=None, want_norm=False):
        """grad_renorm + apply_gradients (main.py:576-577) on self.grads."""
        if lr is None:
            lr = self.lr if self.n_updates_total is None else piecewise_lr(self.n_iters, self.n_updates_total, self.lr)
        norm = ctypes.c_float()
        opt = _lib.JCM_OPT_ADAM if self.optimizer == 'adam' else _lib.JCM_OPT_MOMENTUM
        _lib.check(self._lib.jcm_train_apply(self.eng._h, self.eng._p(self.grads), opt, float(lr), CLIP_NORM,
                                             ctypes.byref(norm) if want_norm else None), 'jcm_train_apply')
        return norm.value if want_norm else None

    def sync_moving_statistics(self, names_and_counts):
        """Keep the replicas identical under data parallelism: every rank advanced moving_mean /
        moving_variance with its own tower's batch statistics (main.py:557 runs the towers' update ops on
        the shared variables one after the other); here the replicas take the mean of the per-rank results,
        i.e. one update with the tower-averaged statistics.  `names_and_counts`: [(name, count)]."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return
        from . import dist as jdist
        if self._moving is None:
            off, lay = 0, []
            for n, c in names_and_counts:
                lay.append((n, off, c))
                off += c
            self._moving = lay
            self._moving_buf = torch.zeros(off, dtype=torch.float32, device=self.eng.device)
        buf = self._moving_buf
        for n, o, c in self._moving:
            _lib.check(self._lib.jcm_get_tensor(self.eng._h, n.encode(), ctypes.c_void_p(buf.data_ptr() + 4 * o), c), 'jcm_get_tensor(%s)' % n)
        jdist.average_gradients(buf, stream=self.eng._stream)
        for i, (n, o, c) in enumerate(self._moving):
            _lib.check(self._lib.jcm_update_tensor(self.eng._h, n.encode(), ctypes.c_void_p(buf.data_ptr() + 4 * o), c,
                                                   int(i == len(self._moving) - 1)), 'jcm_update_tensor(%s)' % n)

    def train_step(self, x, y, want_norm=False, moving=None):
        """One sess.run(train_step) (main.py:644).  Returns (losses tensor, grad norm or None).
        `moving`: [(name, count)] of the BN moving statistics to keep in sync across ranks (N > 1)."""
        self.loss_and_grads(x, y)
        self.average_gradients()
        norm = self.apply(want_norm=want_norm)
        if moving:
            self.sync_moving_statistics(moving)
        return self.losses, norm

    @staticmethod
    def moving_statistics_of(params):
        return [(k, int(np.asarray(v).size)) for k, v in sorted(params.items())
                if k.endswith('moving_mean') or k.endswith('moving_variance')]

    def get_tensor(self, name, shape):
        out = np.empty(int(np.prod(shape)), np.float32)
        _lib.check(self._lib.jcm_get_tensor(self.eng._h, name.encode(), ctypes.c_void_p(out.ctypes.data), out.size),
                   'jcm_get_tensor(%s)' % name)
        return out.reshape(shape)

    def get_params(self, like):
        """All stored parameters with the shapes of the dict `like` (what Saver.save would write)."""
        return {k: self.get_tensor(k, np.asarray(v).shape) for k, v in like.items()}
