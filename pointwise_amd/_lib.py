"""ctypes binding of libconv3p_hip.so -- the C ABI declared in include/conv3p.h.

The product path has no CPU fallback: if the library is missing or a symbol is absent this
module raises, loudly.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CONV3P_HIP_LIB") or os.path.join(_HERE, "csrc", "libconv3p_hip.so")   # env override: developer A/B builds only

OK = 0
ERR_INVALID_ARGUMENT = 1
ERR_WORKSPACE = 2
ERR_UNSUPPORTED = 3
ERR_LAUNCH = 4
ERR_NO_DEVICE = 5

PASS_FORWARD = 0
PASS_BACKWARD = 1
PASS_NEIGHBOR_COUNT = 2
CACHE_POINTS_UNCHANGED = 1
CACHE_SPARSE_NEIGHBOURHOODS = 2   # tuning hints, see include/conv3p.h
CACHE_DENSE_NEIGHBOURHOODS = 8
CACHE_PREPARE_DEEP_ORDERS = 4     # conv3p_cache_prepare_*: also the matrix-core path's record orders
CACHE_FUSED_FORWARD = 16          # conv3p_stack_*: hidden layers of the forward / backward pass as ONE launch (opt-in, see
CACHE_FUSED_BACKWARD = 32         # include/conv3p.h)
CACHE_FUSED_STACK = CACHE_FUSED_FORWARD | CACHE_FUSED_BACKWARD
ABI_VERSION = 5                # CONV3P_ABI_VERSION of include/conv3p.h
STACK_MAX_LAYERS = 8

_vp = ctypes.c_void_p
_i = ctypes.c_int
_sz = ctypes.c_size_t

# every symbol include/conv3p.h declares: name -> (restype, argtypes)
def _sig(real):
    return {
        "forward": (_i, [_vp, _vp, _vp, _vp, real, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
        "backward": (_i, [_vp, _vp, _vp, _vp, _vp, real, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
        "forward_cached": (_i, [_vp, _vp, _vp, _vp, real, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _sz, _vp, _vp]),
        "backward_cached": (_i, [_vp, _vp, _vp, _vp, _vp, real, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp, _vp]),
        "layer_forward_cached": (_i, [_vp, _vp, _vp, _vp, real, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _sz, _vp, _vp]),
        "layer_backward_cached": (_i, [_vp, _vp, _vp, _vp, _vp, real, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _sz,
                                       _vp, _vp]),
        "cache_prepare": (_i, [_vp, _vp, real, _i, _i, _i, _i, _i, _vp, _sz, _vp, _vp]),
        "cache_prepare_multi": (_i, [_vp, _vp, _i, real, _i, _i, _i, _i, _i, _vp, _sz, _vp, _vp]),
        "neighbor_count": (_i, [_vp, _vp, real, _i, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
        "selu": (_i, [_vp, _vp, _sz, _vp]),
        "selu_grad": (_i, [_vp, _vp, _vp, _sz, _vp]),
        "selu_grad_add": (_i, [_vp, _vp, _vp, _vp, _sz, _vp]),
        "stack_prefetch": (_i, [ctypes.POINTER(StackDesc), _vp, real, _i, _i, _vp, _sz, _vp, _vp, _vp]),
        "stack_forward": (_i, [ctypes.POINTER(StackDesc), _vp, _vp, _vp, real, _i, _i, _vp, _vp, _vp, _sz, _vp, _vp, _vp]),
        "stack_backward": (_i, [ctypes.POINTER(StackDesc), _vp, _vp, _vp, real, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                _sz, _vp, _sz, _vp, _vp]),
    }


class CacheConfig(ctypes.Structure):
    """conv3p_cache_config of include/conv3p.h."""
    _fields_ = [("slots", _i), ("max_taps", _i), ("pairs_per_point", _i), ("max_Cin", _i), ("max_Cout", _i),
                ("flags", _i)]


class StackDesc(ctypes.Structure):
    """conv3p_stack_desc of include/conv3p.h."""
    _fields_ = [("n_hidden", _i), ("in_channels", _i), ("hidden", _i), ("num_class", _i), ("fz", _i), ("fy", _i),
                ("fx", _i), ("strides", (ctypes.c_int32 * 3) * (STACK_MAX_LAYERS + 1))]


SYMBOLS = {
    "conv3p_fc_workspace_bytes": (_sz, [_i, _i, _i]),
    "conv3p_fc_forward_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "conv3p_fc_backward_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "conv3p_augment_f32": (_i, [_vp, _vp, _vp, ctypes.c_double, ctypes.c_double, _i, _i, _vp, _vp]),
    "conv3p_sort_xyz_order_f32": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "conv3p_gather_rows": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "conv3p_stack_scratch_bytes": (_sz, [ctypes.POINTER(StackDesc), _i, _i, _i]),
    "conv3p_workspace_bytes": (_sz, [_i] * 9),
    "conv3p_cache_bytes": (_sz, [_i, _i, _i, ctypes.POINTER(CacheConfig)]),
    "conv3p_cache_forget": (_i, [_vp]),
    "conv3p_cache_init": (_i, [_vp, _sz, _vp]),
    "conv3p_cache_fused_status": (_i, [_vp, ctypes.POINTER(ctypes.c_uint), ctypes.POINTER(ctypes.c_uint), ctypes.POINTER(ctypes.c_uint)]),
    "conv3p_profile_enable": (_i, [_i]),
    "conv3p_profile_reset": (_i, []),
    "conv3p_profile_kinds": (_i, []),
    "conv3p_profile_name": (ctypes.c_char_p, [_i]),
    "conv3p_profile_read": (_i, [_i, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_double)]),
    "conv3p_status_string": (ctypes.c_char_p, [_i]),
    "conv3p_abi_version": (_i, []),
}
for _sfx, _real in (("f32", ctypes.c_float), ("f64", ctypes.c_double)):
    for _name, _s in _sig(_real).items():
        SYMBOLS["conv3p_%s_%s" % (_name, _sfx)] = _s

_LIB = None


class Conv3pLibraryError(RuntimeError):
    pass


def load():
    """Load the HIP library and bind every declared symbol (raises if anything is missing)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise Conv3pLibraryError(
            "libconv3p_hip.so not found at %s -- run `python -m pointwise_amd.build` "
            "(there is no CPU fallback in the product path)" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise Conv3pLibraryError("libconv3p_hip.so does not export %s" % name)
        fn.restype = res
        fn.argtypes = args
    if lib.conv3p_abi_version() != ABI_VERSION:
        raise Conv3pLibraryError("libconv3p_hip.so ABI version mismatch")
    _LIB = lib
    return lib


def status_string(code):
    return load().conv3p_status_string(code).decode()
