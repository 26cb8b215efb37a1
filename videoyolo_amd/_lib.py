"""ctypes binding of libvyolo.so (include/vyolo.h).  There is no CPU fallback: if the HIP
library is missing or fails to load, every entry point raises."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvyolo.so")

VY_MAX_TOPK = 1024
VY_CONV_EXACT_FP32, VY_CONV_SPLIT_BF16X3, VY_CONV_SPLIT_BF16X3_TRAIN = 0, 1, 2


class VyError(RuntimeError):
    """A libvyolo entry point returned a negative status."""

    def __init__(self, code, msg):
        super().__init__("libvyolo error %d: %s" % (code, msg))
        self.code = code


class ParamInfo(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 96), ("kind", ctypes.c_int32), ("ndim", ctypes.c_int32),
                ("shape", ctypes.c_int32 * 4), ("size", ctypes.c_int64), ("offset", ctypes.c_int64),
                ("trainable", ctypes.c_int32), ("backbone", ctypes.c_int32)]


class ConvInfo(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 96), ("cin", ctypes.c_int32), ("cout", ctypes.c_int32), ("kernel", ctypes.c_int32),
                ("stride", ctypes.c_int32), ("pad", ctypes.c_int32), ("has_bn", ctypes.c_int32), ("sync_bn", ctypes.c_int32),
                ("residual", ctypes.c_int32), ("upsample", ctypes.c_int32), ("concat_offset", ctypes.c_int32),
                ("out_channels_total", ctypes.c_int32)]


class LaunchStat(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 64), ("ms", ctypes.c_float), ("flops", ctypes.c_double),
                ("bytes", ctypes.c_double)]


_vp, _i32, _f32, _sz = ctypes.c_void_p, ctypes.c_int32, ctypes.c_float, ctypes.c_size_t

# name -> (restype, argtypes); the single source of truth for the exported-symbols test
SIGNATURES = {
    "vy_last_error": (ctypes.c_char_p, []),
    "vy_version": (ctypes.c_char_p, []),
    "vy_net_create": (ctypes.c_int, [_i32, ctypes.POINTER(_vp)]),
    "vy_net_destroy": (None, [_vp]),
    "vy_net_set_nms": (ctypes.c_int, [_vp, _f32, _i32, _i32]),
    "vy_net_num_params": (_i32, [_vp]),
    "vy_net_param_info": (ctypes.c_int, [_vp, _i32, ctypes.POINTER(ParamInfo)]),
    "vy_net_num_convs": (_i32, [_vp]),
    "vy_net_conv_info": (ctypes.c_int, [_vp, _i32, ctypes.POINTER(ConvInfo)]),
    "vy_net_param_bytes": (_sz, [_vp]),
    "vy_net_bind_params": (ctypes.c_int, [_vp, _vp]),
    "vy_net_param_set": (ctypes.c_int, [_vp, _i32, _vp, _vp]),
    "vy_net_param_get": (ctypes.c_int, [_vp, _i32, _vp, _vp]),
    "vy_net_workspace_bytes": (_sz, [_vp, _i32, _i32, _i32]),
    "vy_net_bind_workspace": (ctypes.c_int, [_vp, _vp, _sz, _i32, _i32, _i32, _vp]),
    "vy_net_set_keep_activations": (ctypes.c_int, [_vp, _i32]),
    "vy_net_set_conv_mode": (ctypes.c_int, [_vp, _i32]),
    "vy_net_get_conv_mode": (_i32, [_vp]),
    "vy_net_invalidate_split_weights": (ctypes.c_int, [_vp]),
    "vy_net_streamk_state": (ctypes.c_int, [_vp, ctypes.POINTER(_i32), ctypes.POINTER(_sz), ctypes.POINTER(_i32)]),
    "vy_net_num_anchors": (_i32, [_vp]),
    "vy_net_forward_infer": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vy_net_detect_heads": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vy_net_read_head": (ctypes.c_int, [_vp, _i32, _vp, _vp]),
    "vy_net_read_activation": (ctypes.c_int, [_vp, ctypes.c_char_p, _vp, ctypes.POINTER(_i32),
                                              ctypes.POINTER(_i32), ctypes.POINTER(_i32), _vp]),
    "vy_net_profile_infer": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, ctypes.POINTER(LaunchStat),
                                            ctypes.POINTER(_i32), _vp]),
}

ALLREDUCE_CB = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64)
GRAD_BUCKET_CB = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64)

SIGNATURES.update({
    "vy_stream_create": (ctypes.c_int, [ctypes.POINTER(_vp)]),
    "vy_stream_destroy": (ctypes.c_int, [_vp]),
    "vy_prefetch_targets": (ctypes.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vy_preprocess_frames": (ctypes.c_int, [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp]),
    "vy_preprocess_resize_frames": (ctypes.c_int, [_vp, _i32, _i32, _vp, _i32, _i32, _i32, _vp, _vp, _vp]),
    "vy_net_train_workspace_bytes": (_sz, [_vp, _i32, _i32, _i32]),
    "vy_net_bind_train": (ctypes.c_int, [_vp, _vp, _sz, _i32, _i32, _i32, _vp, _vp, _vp]),
    "vy_net_set_train_options": (ctypes.c_int, [_vp, _f32, _i32]),
    "vy_net_train_forward": (ctypes.c_int, [_vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vy_net_train_mode_forward": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vy_net_train_backward": (ctypes.c_int, [_vp, _vp, _vp]),
    "vy_net_param_set_opt": (ctypes.c_int, [_vp, _i32, _f32, _f32, _i32]),
    "vy_net_sgd_step": (ctypes.c_int, [_vp, _f32, _f32, _f32, _f32, _vp]),
    "vy_net_grad_get": (ctypes.c_int, [_vp, _i32, _vp, _vp]),
    "vy_net_read_grad_activation": (ctypes.c_int, [_vp, ctypes.c_char_p, _vp, _vp]),
    "vy_net_set_sync_bn": (ctypes.c_int, [_vp, _i32, ALLREDUCE_CB, _vp]),
    "vy_net_set_grad_bucket_cb": (ctypes.c_int, [_vp, GRAD_BUCKET_CB, _vp]),
})

_lib = None


def load():
    """Load libvyolo.so (built by ``python -m videoyolo_amd.build`` / ``__graft_entry__.build()``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s not found: build the HIP library first (python -m videoyolo_amd.build). "
            "videoyolo_amd has no CPU fallback." % LIB_PATH)
    # torch ships its own libamdhip64; it must be the one HIP runtime of the process, so it is
    # loaded first and libvyolo.so binds to it (loading /opt/rocm's copy first leaves torch without
    # a device).  torch is the plumbing for device memory / streams anyway.
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise VyError(rc, load().vy_last_error().decode())
