"""Handle object over the C ABI: one Engine per (device, stream) holds the packed weights,
spatial-model tables and workspace that the reference keeps as module globals + a
tf.Session (SURVEY.md 8b 'Implicit state').  torch is used only to own device buffers."""
import ctypes

import numpy as np
import torch

from . import _lib


def _hm_size(n):
    """Three SAME stride-2 stages (conv1 + two max-pools): ceil(ceil(ceil(n/2)/2)/2)."""
    for _ in range(3):
        n = (n + 1) // 2
    return n


class Engine:
    """Owns a jcm_handle.  All tensor arguments are torch CUDA float32 NHWC, contiguous."""

    def __init__(self, device=0, precision='fp32', n_joints=9, stream=None, f32_conv=None, split_min_wgs=None, micro_batch=None, conv9_fft=None, call_order=None, fft_single=None, fft_t16=None, fft_fuse=None):
        if not torch.cuda.is_available():
            raise RuntimeError('joint-cnn-mrf_amd needs an MI355X (gfx950) GPU; torch.cuda.is_available() is False '
                               'and there is no CPU path')
        self._lib = _lib.load()
        self.device = torch.device('cuda', device if isinstance(device, int) else torch.device(device).index or 0)
        self.n_joints = int(n_joints)
        self.precision = precision
        self._stream = stream if stream is not None else torch.cuda.current_stream(self.device)
        h = ctypes.c_void_p()
        _lib.check(self._lib.jcm_create(self.device.index, ctypes.c_void_p(self._stream.cuda_stream), ctypes.byref(h)),
                   'jcm_create')
        self._h = h
        self._finalized = False
        prec = {'fp32': _lib.JCM_PRECISION_F32, 'f32': _lib.JCM_PRECISION_F32, 'bf16': _lib.JCM_PRECISION_BF16}[precision]
        _lib.check(self._lib.jcm_set_option(self._h, b'precision', prec), 'jcm_set_option(precision)')
        _lib.check(self._lib.jcm_set_option(self._h, b'n_joints', self.n_joints), 'jcm_set_option(n_joints)')
        if f32_conv is not None:      # 'exact' = the default; 'split16' = the direct kernels on two fp16 parts per operand (fp16x3: fp32-class accuracy on the 16-bit matrix cores)
            if f32_conv not in ('exact', 'split16'):
                raise ValueError("f32_conv must be 'exact' or 'split16' (the bf16x6 arm 'split' was retired in round 5)")
            _lib.check(self._lib.jcm_set_option(self._h, b'f32_conv', {'exact': 0, 'split16': 2}[f32_conv]), 'jcm_set_option(f32_conv)')
        if micro_batch is not None:   # forward() walks a batch in slices of this many images (default 256 bf16 / 64 fp32)
            self.set_micro_batch(micro_batch)
        if conv9_fft is not None:     # False: the wide 9x9 layers of an fp32 engine on the fp32 MFMA chain instead of the frequency domain
            self.set_conv9_fft(conv9_fft)
        if fft_single is not None:    # bf16 engines: False = two bf16 parts per operand of the channel GEMM (three products) instead of one scaled fp16 part
            self.set_option('fft_single', int(bool(fft_single)))
        if fft_t16 is not None:       # bf16 engines: False = the row-transformed tensors and the product spectra of the frequency-domain route stay complex fp32 (default: complex fp16)
            self.set_option('fft_t16', int(bool(fft_t16)))
        if fft_fuse is not None:      # fp32 engines: 0 = separate pool / merge kernels between the frequency-domain layers (A/B arm of the fused hand-overs; default 3)
            self.set_option('fft_fuse', int(fft_fuse))
        if call_order is not None:    # False (debugging): this engine's calls are not ordered against other engines' on the device
            self.set_option('call_order', int(bool(call_order)))
        if split_min_wgs is not None: # 0 forces the split kernels even on grids too small to pay off (parity tests at small batch)
            _lib.check(self._lib.jcm_set_option(self._h, b'split_min_wgs', int(split_min_wgs)), 'jcm_set_option(split_min_wgs)')

    # ------------------------------------------------------------------ lifecycle
    def close(self):
        if getattr(self, '_h', None) is not None and self._h.value:
            self._lib.jcm_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ parameters
    def set_tensor(self, name, value):
        """`name` is the reference's TF variable name; `value` numpy or torch (host or device)."""
        if isinstance(value, torch.Tensor):
            t = value.detach().to(torch.float32).contiguous()
            ptr, shape, keep = t.data_ptr(), tuple(t.shape), t
        else:
            a = np.ascontiguousarray(value, dtype=np.float32)
            ptr, shape, keep = a.ctypes.data, a.shape, a
        if len(shape) == 0:
            raise ValueError('scalar parameter %r' % name)
        arr = (ctypes.c_int64 * len(shape))(*shape)
        _lib.check(self._lib.jcm_set_tensor(self._h, name.encode(), ctypes.c_void_p(ptr), arr, len(shape)),
                   'jcm_set_tensor(%s)' % name)
        del keep

    def load_params(self, params, finalize=True):
        for name, value in params.items():
            self.set_tensor(name, value)
        if finalize:
            self.finalize()
        return self

    def update_tensor(self, name, value, refresh=True):
        """Overwrite a stored parameter after finalize (Saver.restore on a live session, main.py:612); refresh rebuilds
        the derived tables (pass False on all but the last tensor of a batch of updates)."""
        a = np.ascontiguousarray(value, dtype=np.float32)
        _lib.check(self._lib.jcm_update_tensor(self._h, name.encode(), ctypes.c_void_p(a.ctypes.data), a.size, int(bool(refresh))),
                   'jcm_update_tensor(%s)' % name)

    def finalize(self):
        _lib.check(self._lib.jcm_finalize(self._h), 'jcm_finalize')
        self._finalized = True

    # ------------------------------------------------------------------ helpers
    def _chk(self, t, ndim, name, dtype=torch.float32):
        if not isinstance(t, torch.Tensor):
            raise TypeError('%s must be a torch tensor' % name)
        if t.device != self.device:
            raise ValueError('%s is on %s, engine is on %s' % (name, t.device, self.device))
        if t.dtype != dtype:
            raise TypeError('%s must be %s, got %s' % (name, dtype, t.dtype))
        if t.dim() != ndim:
            raise ValueError('%s must have %d dims (NHWC), got shape %s' % (name, ndim, tuple(t.shape)))
        if not t.is_contiguous():
            raise ValueError('%s must be contiguous (NHWC)' % name)
        return t

    def _new(self, *shape, dtype=torch.float32):
        return torch.empty(shape, dtype=dtype, device=self.device)

    @staticmethod
    def _p(t):
        return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)

    # ------------------------------------------------------------------ ops (main.py names)
    def conv_layer(self, x, name, stride, last_layer=False, n_out=None):
        """main.py:156-169."""
        self._chk(x, 4, 'x')
        B, H, W, _ = x.shape
        if n_out is None:
            raise ValueError('n_out is required to size the output')
        out = self._new(B, -(-H // stride), -(-W // stride), n_out)
        _lib.check(self._lib.jcm_conv_layer(self._h, name.encode(), stride, int(bool(last_layer)), self._p(x), B, H, W,
                                            self._p(out)), 'jcm_conv_layer(%s)' % name)
        return out

    def conv_layer_merged(self, x1, x2, x3, name, n_out):
        """conv_layer(((x1 + up(x2)) + up(x3)) / 3) (main.py:58,67,69-71), the merge formed as the tower forms it (inside the layer's forward row
        pass on the frequency-domain route)."""
        for t, nm in ((x1, 'x1'), (x2, 'x2'), (x3, 'x3')):
            self._chk(t, 4, nm)
        B, H, W, C = x1.shape
        if x2.shape[0] != B or x3.shape[0] != B or x2.shape[3] != C or x3.shape[3] != C:
            raise ValueError('x1, x2, x3 must share batch and channel counts')
        out = self._new(B, H, W, n_out)
        _lib.check(self._lib.jcm_conv_layer_merged(self._h, name.encode(), self._p(x1), self._p(x2), x2.shape[1], x2.shape[2], self._p(x3), x3.shape[1], x3.shape[2],
                                                   B, H, W, self._p(out)), 'jcm_conv_layer_merged(%s)' % name)
        return out

    def max_pool(self, x):
        """main.py:172-174."""
        self._chk(x, 4, 'x')
        B, H, W, C = x.shape
        out = self._new(B, (H + 1) // 2, (W + 1) // 2, C)
        _lib.check(self._lib.jcm_max_pool(self._h, self._p(x), B, H, W, C, self._p(out)), 'jcm_max_pool')
        return out

    def resize_bilinear(self, x, oh, ow):
        """tf.image.resize_images, TF-1.x legacy bilinear (main.py:51,58,60,67,89)."""
        self._chk(x, 4, 'x')
        B, H, W, C = x.shape
        out = self._new(B, oh, ow, C)
        _lib.check(self._lib.jcm_resize_bilinear(self._h, self._p(x), B, H, W, C, oh, ow, self._p(out)), 'jcm_resize_bilinear')
        return out

    def model(self, x):
        """main.py:29-74: [B,H,W,3] -> logits [B,H/8,W/8,K]."""
        self._chk(x, 4, 'x')
        B, H, W, C = x.shape
        if C != 3:
            raise ValueError('x must be [B,H,W,3], got %s' % (tuple(x.shape),))
        out = self._new(B, _hm_size(H), _hm_size(W), self.n_joints)
        _lib.check(self._lib.jcm_pd_forward(self._h, self._p(x), B, H, W, self._p(out)), 'jcm_pd_forward')
        return out

    def spatial_softmax(self, hm):
        """main.py:212-217."""
        self._chk(hm, 4, 'hm')
        B, H, W, K = hm.shape
        out = torch.empty_like(hm)
        _lib.check(self._lib.jcm_spatial_softmax(self._h, self._p(hm), B, H * W, K, self._p(out)), 'jcm_spatial_softmax')
        return out

    def conv_mrf(self, A, Bm):
        """main.py:77-91: A [1,120,180,1], B [b,60,90,1] -> [b,60,90,1]."""
        self._chk(A, 4, 'A')
        self._chk(Bm, 4, 'B')
        if tuple(A.shape) != (1, 120, 180, 1) or tuple(Bm.shape[1:]) != (60, 90, 1):
            raise ValueError('conv_mrf expects A [1,120,180,1] and B [b,60,90,1]; got %s, %s' % (tuple(A.shape), tuple(Bm.shape)))
        out = torch.empty_like(Bm)
        _lib.check(self._lib.jcm_conv_mrf(self._h, self._p(A), self._p(Bm), Bm.shape[0], self._p(out)), 'jcm_conv_mrf')
        return out

    def spatial_model(self, heat_map):
        """main.py:94-125: [B,60,90,K+1] -> [B,60,90,K]."""
        self._chk(heat_map, 4, 'heat_map')
        B = heat_map.shape[0]
        if tuple(heat_map.shape[1:]) != (60, 90, self.n_joints + 1):
            raise ValueError('spatial_model expects [B,60,90,%d], got %s' % (self.n_joints + 1, tuple(heat_map.shape)))
        out = self._new(B, 60, 90, self.n_joints)
        _lib.check(self._lib.jcm_sm_forward(self._h, self._p(heat_map), B, self._p(out)), 'jcm_sm_forward')
        return out

    def argmax_coords(self, hm):
        """evaluation.py:15-24: [B,H,W,K] -> int32 [B,2,K] (row, col)."""
        self._chk(hm, 4, 'hm')
        B, H, W, K = hm.shape
        out = self._new(B, 2, K, dtype=torch.int32)
        _lib.check(self._lib.jcm_argmax_coords(self._h, self._p(hm), B, H, W, K, self._p(out)), 'jcm_argmax_coords')
        return out

    def softmax_argmax(self, logits, want_prob=True):
        """spatial_softmax + arg-max of the probabilities in one kernel (the tail of forward()):
        [B,H,W,K] logits -> (prob [B,H,W,K] or None, coords int32 [B,2,K])."""
        self._chk(logits, 4, 'logits')
        B, H, W, K = logits.shape
        prob = torch.empty_like(logits) if want_prob else None
        coords = self._new(B, 2, K, dtype=torch.int32)
        _lib.check(self._lib.jcm_softmax_argmax(self._h, self._p(logits), B, H, W, K, self._p(prob), self._p(coords)), 'jcm_softmax_argmax')
        return prob, coords

    def forward(self, x, torso=None, use_sm=True, want_prob=True):
        """The tower of main.py:522-531 in one C call.  Returns a dict with 'pd_coords',
        'sm_coords' (int32 [B,2,K]) and, if want_prob, 'pd_prob' / 'sm_prob' [B,60,90,K]."""
        self._chk(x, 4, 'x')
        B, H, W, C = x.shape
        if C != 3:
            raise ValueError('x must be [B,H,W,3]')
        if use_sm:
            if torso is None:
                raise ValueError('use_sm=True needs the torso heat map y_in[..., 9:] (main.py:528)')
            self._chk(torso, 4, 'torso')
            if tuple(torso.shape) != (B, 60, 90, 1):
                raise ValueError('torso must be [B,60,90,1], got %s' % (tuple(torso.shape),))
        hh, ww, K = _hm_size(H), _hm_size(W), self.n_joints
        r = {'pd_coords': self._new(B, 2, K, dtype=torch.int32)}
        if want_prob:
            r['pd_prob'] = self._new(B, hh, ww, K)
        if use_sm:
            r['sm_coords'] = self._new(B, 2, K, dtype=torch.int32)
            if want_prob:
                r['sm_prob'] = self._new(B, hh, ww, K)
        _lib.check(self._lib.jcm_forward(self._h, self._p(x), self._p(torso if use_sm else None), B, H, W, int(bool(use_sm)),
                                         self._p(r.get('pd_prob')), self._p(r.get('sm_prob')),
                                         self._p(r['pd_coords']), self._p(r.get('sm_coords'))), 'jcm_forward')
        return r

    def eval_forward(self, x, y, use_sm=True, want_prob=True):
        """The tower in inference mode plus the two cross-entropy losses of the graph (main.py:538-539), as eval_error
        runs it per batch (main.py:275-283).  y = y_in [B,60,90,K+1]: targets + torso channel.  Returns the dict of
        forward() plus 'losses' (device fp32 [2]: loss_pd, loss_sm)."""
        self._chk(x, 4, 'x')
        self._chk(y, 4, 'y')
        B, H, W, C = x.shape
        hh, ww, K = _hm_size(H), _hm_size(W), self.n_joints
        if C != 3 or tuple(y.shape) != (B, hh, ww, K + 1):
            raise ValueError('x must be [B,H,W,3] and y [B,%d,%d,%d]; got %s, %s' % (hh, ww, K + 1, tuple(x.shape), tuple(y.shape)))
        r = {'pd_coords': self._new(B, 2, K, dtype=torch.int32), 'losses': self._new(2)}
        if want_prob:
            r['pd_prob'] = self._new(B, hh, ww, K)
        if use_sm:
            r['sm_coords'] = self._new(B, 2, K, dtype=torch.int32)
            if want_prob:
                r['sm_prob'] = self._new(B, hh, ww, K)
        _lib.check(self._lib.jcm_eval_forward(self._h, self._p(x), self._p(y), B, H, W, int(bool(use_sm)),
                                              self._p(r.get('pd_prob')), self._p(r.get('sm_prob')), self._p(r['pd_coords']),
                                              self._p(r.get('sm_coords')), self._p(r['losses'])), 'jcm_eval_forward')
        return r

    def window_resize(self, src, windows, oh, ow):
        """Pad-or-crop `windows` [(src_index, y0, x0, h, w), ...] of src [N,H,W,C], each resized to
        (oh, ow) with skimage.transform.resize's 0.13.x defaults (main.py:326-379)."""
        self._chk(src, 4, 'src')
        N, H, W, C = src.shape
        wins = np.ascontiguousarray(np.asarray(windows, dtype=np.int32).reshape(-1, 5))
        out = self._new(wins.shape[0], int(oh), int(ow), C)
        _lib.check(self._lib.jcm_window_resize(self._h, self._p(src), N, H, W, C,
                                               wins.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), wins.shape[0],
                                               int(oh), int(ow), self._p(out)), 'jcm_window_resize')
        return out

    def group_mean(self, x, group):
        """np.average over consecutive groups of `group` leading entries (main.py:413-414)."""
        if x.shape[0] % group:
            raise ValueError('leading dimension %d is not a multiple of %d' % (x.shape[0], group))
        n = x.shape[0] // group
        out = self._new(n, *x.shape[1:])
        m = 1
        for d in x.shape[1:]:
            m *= int(d)
        _lib.check(self._lib.jcm_group_mean(self._h, self._p(x.contiguous()), n, group, m, self._p(out)), 'jcm_group_mean')
        return out

    def set_option(self, key, value):
        """jcm_set_option(key, value) -- include/jcm.h lists the keys."""
        _lib.check(self._lib.jcm_set_option(self._h, key.encode(), int(value)), 'jcm_set_option(%s)' % key)

    def set_sm_algo(self, algo):
        """Pairwise-convolution algorithm of the spatial model: 'fft_fused' (default; every transform in LDS, sm_fused.hip / sm_lds.hip;
        'fft' is an alias) or 'direct' (LDS sliding-window VALU kernel, the independent cross-check).  Both are hand-written HIP paths; the
        rocFFT routes of rounds 1-4 ('fft', 'fft_split') were removed in round 5."""
        _lib.check(self._lib.jcm_set_option(self._h, b'sm_algo', {'fft': 3, 'fft_fused': 3, 'direct': 1}[algo]), 'jcm_set_option(sm_algo)')

    def set_conv9_fft(self, on):
        """fp32 engines: run the wide 9x9 layers in the frequency domain (in-LDS FFTs + one complex channel GEMM per frequency;
        default) or on the fp32 MFMA accumulation chain.  Both pass the same parity tests."""
        _lib.check(self._lib.jcm_set_option(self._h, b'conv9_fft', int(bool(on))), 'jcm_set_option(conv9_fft)')

    def set_micro_batch(self, n):
        """Images per internal slice of forward(): bounds the workspace when a rank holds a large share of a
        global batch (BASELINE configs[3]: 2048 images over the ranks).  0 = default (256 bf16 / 64 fp32)."""
        _lib.check(self._lib.jcm_set_option(self._h, b'micro_batch', int(n)), 'jcm_set_option(micro_batch)')

    def set_profile(self, on):
        _lib.check(self._lib.jcm_set_option(self._h, b'profile', int(bool(on))), 'jcm_set_option(profile)')

    def profile_read(self, scope):
        """(total_ms, launches) of the HIP-event-bracketed launches of conv layer `scope`."""
        ms, n = ctypes.c_double(0), ctypes.c_int(0)
        _lib.check(self._lib.jcm_profile_read(self._h, scope.encode(), ctypes.byref(ms), ctypes.byref(n)), 'jcm_profile_read')
        return ms.value, n.value

    def conv_kernel_name(self, scope, B, H, W):
        """The HIP kernel a [B,H,W,Cin] launch of conv layer `scope` takes on this engine."""
        buf = ctypes.create_string_buffer(128)
        _lib.check(self._lib.jcm_conv_kernel_name(self._h, scope.encode(), int(B), int(H), int(W), buf, 128), 'jcm_conv_kernel_name')
        return buf.value.decode()

    def workspace_bytes(self):
        return int(self._lib.jcm_workspace_bytes(self._h))
