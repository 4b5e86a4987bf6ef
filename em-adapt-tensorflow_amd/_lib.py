"""ctypes binding of libemadapt_hip.so -- the only way the Python side reaches the kernels.

There is no CPU fallback: if the shared library is missing (and cannot be built because hipcc is
absent) importing the operators fails loudly with ``EmadaptLibraryError``.
"""
import ctypes
import os
from ctypes import POINTER, c_bool, c_double, c_float, c_int, c_int32, c_int64, c_longlong, c_size_t, c_uint64, c_void_p

from . import _build

LIB_PATH = _build.LIB_PATH


class EmadaptLibraryError(ImportError):
    pass


class EmadaptError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        lib = _LIB
        msg = lib.emadapt_strerror(code).decode() if lib is not None else str(code)
        extra = ""
        if lib is not None and code == -3:
            extra = " (hipError_t=%d)" % lib.emadapt_last_hip_error()
        super().__init__("%s failed: %s%s" % (where, msg, extra))


class EmadaptDeviceError(EmadaptError):
    """A kernel recorded a failure in the device status word (include/emadapt.h, emadapt_status_poll): ``flags`` holds its bits
    (STATUS_STREAMK_TIMEOUT ...).  While the word was raised the optimiser kernels skipped their updates."""

    def __init__(self, flags, where):
        super().__init__(-6, where)
        self.flags = flags


STATUS_STREAMK_TIMEOUT = 1
STATUS_NAN_GRADIENT = 2


class VocSample(ctypes.Structure):
    """Mirror of emadapt_voc_sample (include/emadapt.h)."""
    _fields_ = [("img_off", ctypes.c_int64), ("lab_off", ctypes.c_int64), ("h0", ctypes.c_int32), ("w0", ctypes.c_int32),
                ("h1", ctypes.c_int32), ("w1", ctypes.c_int32), ("flip", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class ConvDesc(ctypes.Structure):
    """Mirror of emadapt_conv_desc (include/emadapt.h)."""

    _fields_ = [("B", c_int), ("H", c_int), ("W", c_int), ("Cin", c_int), ("Cout", c_int),
                ("KH", c_int), ("KW", c_int), ("dil", c_int)]


K_NAMES = ["igemm_fwd", "igemm_dgrad", "igemm_wgrad", "estep", "other"]

LABEL_F32, LABEL_I32, LABEL_MULTIHOT = 0, 1, 2
EPI_NONE, EPI_BIAS, EPI_BIAS_RELU = 0, 1, 2

_vp = c_void_p
# name -> (restype, argtypes); status-returning entries are checked by `call`.
PROTOTYPES = {
    "emadapt_version": (c_int, []),
    "emadapt_strerror": (ctypes.c_char_p, [c_int]),
    "emadapt_last_hip_error": (c_int, []),
    "emadapt_device_count": (c_int, []),
    "e_step": (_vp, [_vp, POINTER(c_int), POINTER(c_int), _vp, c_bool, c_int, c_float, c_float, c_float]),
    "e_step_perm": (_vp, [_vp, POINTER(c_int), POINTER(c_int), _vp, _vp, c_bool, c_int, c_float, c_float, c_float]),
    "emadapt_estep_seed": (None, [c_uint64]),
    "emadapt_estep_host": (c_int, [_vp, c_int, c_int, c_int, c_int, _vp, c_int, c_int64, _vp, c_int, c_int, c_float,
                                   c_int, c_int, _vp]),
    "emadapt_estep_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "emadapt_estep_needs_map": (c_int, [c_int, c_int, c_int]),
    "emadapt_estep": (c_int, [_vp, c_int64, _vp, c_int, c_int64, _vp, c_int, c_int, c_int, c_int, c_int, c_int,
                              c_float, c_int, c_int, _vp, _vp, _vp, c_size_t, _vp]),
    "emadapt_pred_bilinear_argmax": (c_int, [_vp, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, _vp, _vp, _vp]),
    "emadapt_bgr_mean_flip": (c_int, [_vp, _vp, c_int, c_int, c_int, POINTER(c_float), _vp, _vp, _vp, _vp]),
    "emadapt_resize_nearest": (c_int, [_vp, c_int, c_int, c_int, _vp, c_int, c_int, _vp]),
    "emadapt_host_register": (c_int, [_vp, c_size_t]),
    "emadapt_host_unregister": (c_int, [_vp]),
    "emadapt_memcpy_h2d_async": (c_int, [_vp, _vp, c_size_t, _vp]),
    "emadapt_voc_preprocess": (c_int, [_vp, c_int64, POINTER(VocSample), c_int, c_int, c_int, POINTER(c_float), _vp, _vp, _vp]),
    "emadapt_conv_fwd_workspace_bytes": (c_size_t, [POINTER(ConvDesc)]),
    "emadapt_conv_fwd": (c_int, [POINTER(ConvDesc), _vp, _vp, _vp, _vp, c_int, c_float, _vp, c_uint64, _vp, c_size_t,
                                 _vp]),
    "emadapt_conv_dgrad_workspace_bytes": (c_size_t, [POINTER(ConvDesc)]),
    "emadapt_conv_dgrad": (c_int, [POINTER(ConvDesc), _vp, _vp, _vp, _vp, c_float, _vp, c_size_t, _vp]),
    "emadapt_conv_wgrad_workspace_bytes": (c_size_t, [POINTER(ConvDesc)]),
    "emadapt_conv_wgrad": (c_int, [POINTER(ConvDesc), _vp, _vp, _vp, _vp, c_size_t, _vp]),
    "emadapt_conv_wgrad_bias_fused": (c_int, [POINTER(ConvDesc)]),
    "emadapt_conv_wgrad_bias_workspace_bytes": (c_size_t, [POINTER(ConvDesc)]),
    "emadapt_conv_wgrad_bias": (c_int, [POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, c_size_t, _vp]),
    "emadapt_filter_kmajor": (c_int, [POINTER(ConvDesc), _vp, _vp, _vp]),
    "emadapt_conv_fwd_x6_workspace_bytes": (c_size_t, [POINTER(ConvDesc)]),
    "emadapt_conv_fwd_x6": (c_int, [POINTER(ConvDesc), _vp, _vp, _vp, _vp, c_int, c_float, _vp, c_uint64, _vp, c_size_t, _vp]),
    "emadapt_conv_dgrad_x6_workspace_bytes": (c_size_t, [POINTER(ConvDesc)]),
    "emadapt_conv_dgrad_x6": (c_int, [POINTER(ConvDesc), _vp, _vp, _vp, _vp, c_float, _vp, c_size_t, _vp]),
    "emadapt_conv_wgrad_x6_workspace_bytes": (c_size_t, [POINTER(ConvDesc)]),
    "emadapt_conv_wgrad_x6": (c_int, [POINTER(ConvDesc), _vp, _vp, _vp, _vp, c_size_t, _vp]),
    "emadapt_planes_bytes": (c_size_t, [c_longlong, c_int]),
    "emadapt_split_planes": (c_int, [_vp, _vp, c_longlong, c_int, _vp]),
    "emadapt_filter_tiles_bytes": (c_size_t, [POINTER(ConvDesc), c_int]),
    "emadapt_filter_tiles": (c_int, [POINTER(ConvDesc), _vp, _vp, c_int, _vp]),
    "emadapt_conv_x6p_supported": (c_int, [POINTER(ConvDesc), c_int]),
    "emadapt_conv_x6p_pays": (c_int, [POINTER(ConvDesc), c_int]),
    "emadapt_conv_x6p_plan": (c_int, [POINTER(ConvDesc), c_int, POINTER(c_int)]),
    "emadapt_conv_fwd_x6p_workspace_bytes": (c_size_t, [POINTER(ConvDesc)]),
    "emadapt_conv_fwd_x6p": (c_int, [POINTER(ConvDesc), _vp, _vp, _vp, _vp, c_int, c_float, _vp, c_uint64, _vp, c_size_t, _vp]),
    "emadapt_conv_dgrad_x6p_workspace_bytes": (c_size_t, [POINTER(ConvDesc)]),
    "emadapt_conv_dgrad_x6p": (c_int, [POINTER(ConvDesc), _vp, _vp, _vp, _vp, c_float, _vp, c_size_t, _vp]),
    "emadapt_conv_wgrad_x6p_workspace_bytes": (c_size_t, [POINTER(ConvDesc)]),
    "emadapt_conv_wgrad_x6p": (c_int, [POINTER(ConvDesc), _vp, _vp, _vp, _vp, c_size_t, _vp]),
    "emadapt_maxpool_fwd_dt": (c_int, [_vp, c_int, c_int, c_int, c_int, c_int, c_int, _vp, _vp, _vp]),
    "emadapt_maxpool_bwd_dt": (c_int, [_vp, _vp, _vp, c_int, c_int, c_int, c_int, c_int, c_int, _vp, _vp]),
    "emadapt_maxpool_fwd_gated": (c_int, [_vp, c_int, c_int, c_int, c_int, c_int, c_int, _vp, _vp, c_int, _vp]),
    "emadapt_bias_grad_dt": (c_int, [_vp, c_int, c_int64, c_int, _vp, _vp, c_size_t, _vp]),
    "emadapt_softmax_ce_dt": (c_int, [_vp, c_int64, _vp, c_int64, c_int, c_float, _vp, c_int, _vp, _vp, c_size_t, _vp]),
    "emadapt_weights_bf16_bytes": (c_size_t, [POINTER(ConvDesc), c_int]),
    "emadapt_weights_to_bf16": (c_int, [POINTER(ConvDesc), _vp, _vp, _vp, _vp]),
    "emadapt_weights_to_bf16_multi": (c_int, [c_int, _vp, _vp, _vp, _vp, _vp]),
    "emadapt_conv_fwd_bf16_workspace_bytes": (c_size_t, [POINTER(ConvDesc)]),
    "emadapt_conv_fwd_bf16": (c_int, [POINTER(ConvDesc), _vp, c_int, _vp, _vp, _vp, c_int, c_int, c_float, _vp, c_uint64,
                                      _vp, c_size_t, _vp]),
    "emadapt_conv_dgrad_bf16_workspace_bytes": (c_size_t, [POINTER(ConvDesc)]),
    "emadapt_conv_dgrad_bf16": (c_int, [POINTER(ConvDesc), _vp, c_int, _vp, _vp, c_int, _vp, c_float, _vp, c_size_t,
                                        _vp]),
    "emadapt_conv_wgrad_bf16_workspace_bytes": (c_size_t, [POINTER(ConvDesc)]),
    "emadapt_conv_wgrad_bf16": (c_int, [POINTER(ConvDesc), _vp, c_int, _vp, c_int, _vp, _vp, c_size_t, _vp]),
    "emadapt_conv_wgrad_bias_bf16_fused": (c_int, [POINTER(ConvDesc), c_int, c_int]),
    "emadapt_conv_wgrad_bias_bf16_workspace_bytes": (c_size_t, [POINTER(ConvDesc)]),
    "emadapt_conv_wgrad_bias_bf16": (c_int, [POINTER(ConvDesc), _vp, c_int, _vp, c_int, _vp, _vp, _vp, c_size_t, _vp]),
    "emadapt_bias_grad_workspace_bytes": (c_size_t, [c_int64, c_int]),
    "emadapt_bias_grad": (c_int, [_vp, c_int64, c_int, _vp, _vp, c_size_t, _vp]),
    "emadapt_maxpool_fwd": (c_int, [_vp, c_int, c_int, c_int, c_int, c_int, _vp, _vp, _vp]),
    "emadapt_maxpool_bwd": (c_int, [_vp, _vp, _vp, c_int, c_int, c_int, c_int, c_int, _vp, _vp]),
    "emadapt_copy_pad_channels": (c_int, [_vp, c_int64, c_int, _vp, c_int, _vp]),
    "emadapt_softmax_ce_workspace_bytes": (c_size_t, [c_int64]),
    "emadapt_softmax_ce": (c_int, [_vp, c_int64, _vp, c_int64, c_int, c_float, _vp, _vp, _vp, c_size_t, _vp]),
    "emadapt_l2_loss_workspace_bytes": (c_size_t, [c_int64]),
    "emadapt_l2_loss": (c_int, [_vp, c_int64, _vp, _vp, c_size_t, _vp]),
    "emadapt_sgd_accumulate": (c_int, [_vp, _vp, _vp, c_int64, c_int64, c_float, c_float, _vp]),
    "emadapt_sgd_momentum": (c_int, [_vp, _vp, _vp, c_int64, c_float, c_float, c_float, _vp]),
    "emadapt_sgd_accumulate_momentum": (c_int, [_vp, _vp, _vp, c_int64, c_int64, c_float, c_float, c_float, c_float, c_float, _vp]),
    "emadapt_sgd_accumulate_momentum_multi": (c_int, [_vp, _vp, _vp, c_int, _vp, _vp, _vp, c_float, c_float, c_float, c_float, c_float, _vp]),
    "emadapt_sgd_momentum_payload_tiles": (c_int, [_vp, _vp, _vp, _vp, _vp, c_int, c_int64, c_float, c_float, c_float, _vp, _vp, _vp]),
    "emadapt_sgd_accumulate_momentum_tiles_supported": (c_int, [_vp]),
    "emadapt_sgd_accumulate_momentum_tiles": (c_int, [_vp, _vp, _vp, _vp, c_int64, c_float, c_float, c_float, c_float, c_float, _vp, _vp, _vp]),
    "emadapt_sgd_momentum_payload": (c_int, [_vp, _vp, _vp, _vp, c_int, c_int64, c_float, c_float, c_float, _vp]),
    "emadapt_sgd_accumulate_pack": (c_int, [_vp, _vp, c_int64, c_int64, c_float, c_float, _vp, c_int, _vp]),
    "emadapt_grad_pack": (c_int, [_vp, c_int64, _vp, c_int, c_int64, _vp]),
    "emadapt_grad_owner_sum": (c_int, [_vp, c_int, c_int, c_int64, _vp, _vp]),
    "emadapt_status_word": (c_int, [POINTER(_vp)]),
    "emadapt_status_poll": (c_int, [POINTER(ctypes.c_uint), c_int, _vp]),
    "emadapt_status_collective": (c_int, [c_int]),
    "emadapt_status_raise": (c_int, [ctypes.c_uint, _vp]),
    "emadapt_conv_plan": (c_int, [POINTER(ConvDesc), c_int, POINTER(c_int)]),
    "emadapt_tune_set": (c_int, [ctypes.c_char_p, c_int]),
    "emadapt_tune_get": (c_int, [ctypes.c_char_p, POINTER(c_int)]),
    "emadapt_prof_enable": (None, [c_int]),
    "emadapt_prof_reset": (None, []),
    "emadapt_prof_collect": (c_int, [POINTER(c_int64), POINTER(c_double), POINTER(c_double), POINTER(c_double)]),
    "emadapt_conv_executed_fraction": (c_int, [POINTER(ConvDesc), c_int, c_int, POINTER(c_double)]),
    "emadapt_conv_bf16_plan": (c_int, [POINTER(ConvDesc), c_int, POINTER(c_int)]),
    "emadapt_conv_bf16t_supported": (c_int, [POINTER(ConvDesc), c_int, c_int]),
    "emadapt_conv_bf16t_pays": (c_int, [POINTER(ConvDesc), c_int, c_int]),
    "emadapt_filter_tiles_bf16_bytes": (c_size_t, [POINTER(ConvDesc), c_int]),
    "emadapt_filter_tiles_bf16": (c_int, [POINTER(ConvDesc), _vp, _vp, c_int, _vp]),
    "emadapt_filter_tiles_bf16_both": (c_int, [POINTER(ConvDesc), _vp, _vp, _vp, _vp]),
    "emadapt_conv_bf16t_workspace_bytes": (c_size_t, [POINTER(ConvDesc), c_int]),
    "emadapt_conv_bf16t_plan": (c_int, [POINTER(ConvDesc), c_int, POINTER(c_int)]),
    "emadapt_conv_fwd_bf16t": (c_int, [POINTER(ConvDesc), _vp, _vp, _vp, _vp, c_int, c_int, c_float, _vp, c_uint64, _vp, c_size_t, _vp]),
    "emadapt_conv_dgrad_bf16t": (c_int, [POINTER(ConvDesc), _vp, _vp, _vp, c_int, _vp, c_float, _vp, c_size_t, _vp]),
    "emadapt_prof_collect2": (c_int, [POINTER(c_int64), POINTER(c_double), POINTER(c_double), POINTER(c_double), POINTER(c_double)]),
}

_LIB = None


def load(build_if_missing=True):
    """dlopen libemadapt_hip.so (building it first when stale/missing and hipcc is available)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # torch ships its own libamdhip64; import it FIRST so that our library binds to the HIP runtime torch
    # already loaded (one runtime per process: torch's streams and allocations are then valid in our launches).
    import torch  # noqa: F401
    if build_if_missing and os.environ.get("EMADAPT_NO_BUILD") != "1":
        try:
            if _build.is_stale():
                _build.build(verbose=False)
        except Exception as e:  # noqa: BLE001 - reported below if the library is really unusable
            if not os.path.exists(LIB_PATH):
                raise EmadaptLibraryError(
                    "libemadapt_hip.so is missing and could not be built (%s). The HIP extension is required: "
                    "there is no CPU fallback." % e) from e
    if not os.path.exists(LIB_PATH):
        raise EmadaptLibraryError("libemadapt_hip.so not found at %s; run `python __graft_entry__.py` / build()" % LIB_PATH)
    return _bind(LIB_PATH)


def _bind(path):
    """dlopen `path` and bind the prototypes.  (scripts/ bind the -DEMADAPT_ABLATE_BUILD library this way, before any
    operator is used; the product path only ever binds LIB_PATH.)"""
    global _LIB
    try:
        lib = ctypes.CDLL(path)
    except OSError as e:
        raise EmadaptLibraryError("cannot load %s: %s" % (path, e)) from e
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise EmadaptLibraryError("libemadapt_hip.so does not export %s" % name) from e
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib


def call(name, *args):
    """Call a status-returning entry point and raise EmadaptError on a non-zero code."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise EmadaptError(rc, name)
    return rc
