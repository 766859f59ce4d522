"""ctypes binding of libjcm.so (include/jcm.h).  There is no fallback: if the HIP library
is missing or a call fails, this raises -- the product path never routes around the kernels."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('JCM_LIB') or os.path.join(_HERE, 'libjcm.so')   # JCM_LIB: alternate build (kernel A/B experiments)

JCM_PRECISION_F32 = 0
JCM_PRECISION_BF16 = 1
JCM_OPT_ADAM = 0
JCM_OPT_MOMENTUM = 1

_c_float_p = ctypes.c_void_p      # device pointers travel as integers
_c_i32_p = ctypes.c_void_p
_handle = ctypes.c_void_p
GRAD_READY_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64)   # jcm_grad_ready_fn

# name -> (restype, argtypes); exactly the declarations of include/jcm.h
SIGNATURES = {
    'jcm_create': (ctypes.c_int, [ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(_handle)]),
    'jcm_destroy': (ctypes.c_int, [_handle]),
    'jcm_last_error': (ctypes.c_char_p, []),
    'jcm_abi_version': (ctypes.c_int, []),
    'jcm_set_option': (ctypes.c_int, [_handle, ctypes.c_char_p, ctypes.c_int64]),
    'jcm_set_tensor': (ctypes.c_int, [_handle, ctypes.c_char_p, _c_float_p, ctypes.POINTER(ctypes.c_int64), ctypes.c_int]),
    'jcm_finalize': (ctypes.c_int, [_handle]),
    'jcm_conv_layer': (ctypes.c_int, [_handle, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, _c_float_p,
                                      ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_float_p]),
    'jcm_conv_layer_merged': (ctypes.c_int, [_handle, ctypes.c_char_p, _c_float_p, _c_float_p, ctypes.c_int, ctypes.c_int, _c_float_p, ctypes.c_int, ctypes.c_int,
                                             ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_float_p]),
    'jcm_max_pool': (ctypes.c_int, [_handle, _c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_float_p]),
    'jcm_resize_bilinear': (ctypes.c_int, [_handle, _c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                           ctypes.c_int, ctypes.c_int, _c_float_p]),
    'jcm_pd_forward': (ctypes.c_int, [_handle, _c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_float_p]),
    'jcm_spatial_softmax': (ctypes.c_int, [_handle, _c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_float_p]),
    'jcm_conv_mrf': (ctypes.c_int, [_handle, _c_float_p, _c_float_p, ctypes.c_int, _c_float_p]),
    'jcm_sm_forward': (ctypes.c_int, [_handle, _c_float_p, ctypes.c_int, _c_float_p]),
    'jcm_argmax_coords': (ctypes.c_int, [_handle, _c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_i32_p]),
    'jcm_softmax_argmax': (ctypes.c_int, [_handle, _c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_float_p, _c_i32_p]),
    'jcm_forward': (ctypes.c_int, [_handle, _c_float_p, _c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                   _c_float_p, _c_float_p, _c_i32_p, _c_i32_p]),
    'jcm_eval_forward': (ctypes.c_int, [_handle, _c_float_p, _c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                        _c_float_p, _c_float_p, _c_i32_p, _c_i32_p, _c_float_p]),
    'jcm_window_resize': (ctypes.c_int, [_handle, _c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                         ctypes.POINTER(ctypes.c_int32), ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_float_p]),
    'jcm_group_mean': (ctypes.c_int, [_handle, _c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int64, _c_float_p]),
    'jcm_profile_read': (ctypes.c_int, [_handle, ctypes.c_char_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)]),
    'jcm_comm_unique_id': (ctypes.c_int, [ctypes.c_char_p]),
    'jcm_comm_create': (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]),
    'jcm_comm_destroy': (ctypes.c_int, [ctypes.c_void_p]),
    'jcm_allgather_coords': (ctypes.c_int, [_handle, ctypes.c_void_p, _c_i32_p, ctypes.c_int, _c_i32_p]),
    'jcm_crc32c': (ctypes.c_uint32, [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32]),
    'jcm_conv_kernel_name': (ctypes.c_int, [_handle, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_int]),
    'jcm_workspace_bytes': (ctypes.c_int64, [_handle]),
    'jcm_train_begin': (ctypes.c_int, [_handle]),
    'jcm_train_param_count': (ctypes.c_int, [_handle, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]),
    'jcm_train_param_info': (ctypes.c_int, [_handle, ctypes.c_int64, ctypes.c_char_p, ctypes.c_int,
                                            ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]),
    'jcm_train_loss_grads': (ctypes.c_int, [_handle, _c_float_p, _c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                            ctypes.c_float, _c_float_p, _c_float_p]),
    'jcm_train_layer_grads': (ctypes.c_int, [_handle, ctypes.c_char_p, _c_float_p, _c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                             _c_float_p, _c_float_p]),
    'jcm_train_apply': (ctypes.c_int, [_handle, _c_float_p, ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.POINTER(ctypes.c_float)]),
    'jcm_train_set_grad_callback': (ctypes.c_int, [_handle, ctypes.c_void_p, ctypes.c_void_p]),
    'jcm_train_steps': (ctypes.c_int, [_handle, ctypes.POINTER(ctypes.c_int64)]),
    'jcm_train_get_state': (ctypes.c_int, [_handle, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_int64)]),
    'jcm_train_set_state': (ctypes.c_int, [_handle, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64]),
    'jcm_get_tensor': (ctypes.c_int, [_handle, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int64]),
    'jcm_update_tensor': (ctypes.c_int, [_handle, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int]),
}

_lib = None


def load():
    """Load libjcm.so (once) and attach the jcm.h prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH) and not os.environ.get('JCM_LIB'):
        # A source-only checkout (the .so is git-ignored): compile the HIP library in place.  This
        # builds the product, it is not a fallback -- without hipcc the load still fails below.
        hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
        if os.path.exists(hipcc):
            import subprocess
            subprocess.call(['make', '-C', os.path.join(_HERE, 'csrc'), '-j8'],
                            env=dict(os.environ, HIPCC=hipcc), stdout=subprocess.DEVNULL)
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            'libjcm.so is not built (%s). Run `python -c "import __graft_entry__ as g; g.build()"` or '
            '`make -C joint-cnn-mrf_amd/csrc`; there is no CPU fallback for this path.' % LIB_PATH)
    # One HIP runtime per process: torch bundles its own libamdhip64.so (SONAME libamdhip64.so.7,
    # same as /opt/rocm's).  Importing torch first makes libjcm's NEEDED entry bind to the copy
    # torch already loaded; the other order loads two runtimes and the second sees no device.
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError here = header/library drift
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error():
    msg = load().jcm_last_error()
    return msg.decode('utf-8', 'replace') if msg else ''


def check(status, what):
    """int status + thread-local message -> RuntimeError (SURVEY.md 8b 'Errors')."""
    if status != 0:
        raise RuntimeError('%s failed (status %d): %s' % (what, status, last_error()))
