"""ctypes binding of `libi2v_hip.so` (C ABI: `include/i2v_hip.h`).

The product path has NO fallback: `load()` raises if the HIP library is missing or is not the
gfx950 build.  (`bind()` only attaches prototypes to an already opened library; the planner unit
tests use it for their host simulation of the ABI.)
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libi2v_hip.so")
HIP_BACKEND = b"hip:gfx950"


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("src", "dst", "cin", "cout", "kh", "kw", "stride", "pad", "relu", "residual")]


class PoolDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("src", "dst", "k", "stride", "pad")]


class Conv3dDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("src", "dst", "cin", "cout", "kt", "kh", "kw", "stride_t", "stride", "pad_t", "pad", "dil_t",
                 "relu", "residual")]


class AttnDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("theta", "phi", "g", "dst")] + [("scale", C.c_float)]


class Pool3dDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("src", "dst", "kt", "k", "stride_t", "stride", "pad_t", "pad")]


class I2VError(RuntimeError):
    pass


_P, _I, _L, _F = C.c_void_p, C.c_int, C.c_int64, C.c_float
_PROTOS = {
    "i2v_create": ([_I, C.POINTER(_P)], _I),
    "i2v_destroy": ([_P], _I),
    "i2v_last_error": ([], C.c_char_p),
    "i2v_abi_version": ([], _I),
    "i2v_backend": ([], C.c_char_p),
    "i2v_backend_stat": ([C.c_char_p], C.c_longlong),
    "i2v_net_create": ([_P, C.POINTER(_I)], _I),
    "i2v_net_destroy": ([_P, _I], _I),
    "i2v_net_add_buffer": ([_P, _I, _I, _I, _I, C.POINTER(_I)], _I),
    "i2v_net_add_tensor": ([_P, _I, _I, _I, _I, _I, C.POINTER(_I)], _I),
    "i2v_net_set_input": ([_P, _I, _I], _I),
    "i2v_net_set_relu_gain": ([_P, _I, _I, _F], _I),
    "i2v_net_add_conv": ([_P, _I, C.POINTER(ConvDesc), _P, _P, _P], _I),
    "i2v_net_add_maxpool": ([_P, _I, C.POINTER(PoolDesc)], _I),
    "i2v_net_add_conv_preact": ([_P, _I, C.POINTER(ConvDesc), _P, _P, _P, _P, _P], _I),
    "i2v_net_add_avgpool": ([_P, _I, C.POINTER(PoolDesc)], _I),
    "i2v_net_add_buffer3d": ([_P, _I, _I, _I, _I, _I, C.POINTER(_I)], _I),
    "i2v_net_add_conv3d": ([_P, _I, C.POINTER(Conv3dDesc), _P, _P, _P], _I),
    "i2v_net_add_maxpool3d": ([_P, _I, C.POINTER(Pool3dDesc)], _I),
    "i2v_net_add_attention": ([_P, _I, C.POINTER(AttnDesc)], _I),
    "i2v_net_tensor_frames": ([_P, _I, _I, C.POINTER(_I)], _I),
    "i2v_net_plan": ([_P, _I, C.POINTER(_I), _I, _I], _I),
    "i2v_net_workspace_bytes": ([_P, _I], C.c_size_t),
    "i2v_net_fusion_info": ([_P, _I, _P], _I),
    "i2v_net_forward": ([_P, _I, _P, _I, _P], _I),
    "i2v_net_hook_info": ([_P, _I, _I, C.POINTER(_P), C.POINTER(_L), C.POINTER(_P), C.POINTER(_L),
                           C.POINTER(_L), C.POINTER(C.c_int32)], _I),
    "i2v_net_backward": ([_P, _I, _P, _I, _P], _I),
    "i2v_net_read_tensor": ([_P, _I, _I, _I, _P, _I, _P], _I),
    "i2v_timing_enable": ([_P, _I], _I),
    "i2v_timing_collect": ([_P, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(_L), _I], _I),
    "i2v_timing_collect_ex": ([_P, C.POINTER(C.c_double), _I, _I], _I),
    "i2v_clip_from_u8_f32": ([_P, _P, _I, _I, _I, _I, _P], _I),
    "i2v_clip_resize_crop_u8_f32": ([_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P], _I),
    "i2v_frames_from_video_f32": ([_P, _P, _P, _I, _I, _I, _I, _P], _I),
    "i2v_compose_f32": ([_P, _P, _P, _I, _I, _I, _I, _F, _I, _P], _I),
    "i2v_cossim_scratch_bytes": ([_L, _I], C.c_size_t),
    "i2v_cossim_fwd_bwd_f32": ([_P, _L, _P, _L, _L, _I, _P, _I, _F, _I, _I, _P, _P, _L, _P, _P], _I),
    "i2v_std_fwd_bwd_f32": ([_P, _L, _L, _I, _I, _I, _P, _P, _L, _P, _P], _I),
    "i2v_std_reduce_f32": ([_P, _L, _L, _I, _P, _P], _I),
    "i2v_std_grad_f32": ([_P, _L, _L, _I, _L, _I, _I, _P, _P, _L, _P, _P], _I),
    "i2v_adam_step_f32": ([_P, _P, _P, _P, _P, _L, _I, _F, C.c_double, C.c_double, C.c_double, C.c_double, _I, _P], _I),
    "i2v_sign_step_f32": ([_P, _P, _P, _L, _L, _F, _F, _P], _I),
    "i2v_sign_step_delta_f32": ([_P, _P, _L, _F, _P], _I),
    "i2v_sign_step_delta_gx_f32": ([_P, _P, _P, _L, _F, _F, _P], _I),
    "i2v_ilaf_reduce_f32": ([_P, _L, _P, _P, _L, _I, _P, _P], _I),
    "i2v_ilaf_grad_f32": ([_P, _L, _P, _P, _L, _I, C.c_double, _I, _I, _P, _P, _L, _P, _P], _I),
    "i2v_ilaf_scratch_bytes": ([_L, _I, _I], C.c_size_t),
    "i2v_ilaf_reduce_seg_f32": ([_P, _L, _P, _P, _L, _I, _I, _P, _P], _I),
    "i2v_ilaf_grad_seg_f32": ([_P, _L, _P, _P, _L, _I, _I, _P, _I, _I, _P, _P, _L, _P, _P], _I),
    "i2v_tap_distance_f32": ([_P, _L, _P, _L, _I, _I, C.c_double, _I, _I, _P, _P, _L, _P, _P], _I),
    "i2v_head_scratch_bytes": ([_I, _I], C.c_size_t),
    "i2v_head_ce_f32": ([_P, _L, _I, _I, _I, _I, _P, _P, _I, _P, _F, _I, _I, _P, _P, _P, _L, _P, _P], _I),
    "i2v_clip_resample_crop_u8_f32": ([_P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P], _I),
    "i2v_tt_grad_mix_f32": ([_P, _P, _P, _P, _I, _L, _I, _I, _F, _P], _I),
    "i2v_resample_nearest_f32": ([_P, _P, _L, _I, _I, _I, _I, _P, _P, _P], _I),
    "i2v_resample_nearest_bwd_f32": ([_P, _P, _L, _I, _I, _I, _I, _P, _P, _P, _P, _P], _I),
    "i2v_dwconv1d_f32": ([_P, _P, _L, _I, _L, _P, _I, _P], _I),
    "i2v_tap_scratch_bytes": ([_L], _L),
    "i2v_tap_perts_f32": ([_P, _P, _P, _I, _I, _I, _I, _I, _P], _I),
    "i2v_tap_sign_abs_f32": ([_P, _P, _P, _L, _P, _P], _I),
    "i2v_tap_grad_f32": ([_P, _P, _P, _I, _I, _I, _I, _I, _F, _P], _I),
    "i2v_grad_post_scratch_bytes": ([_I, _I, _I, _I, _I, _I], _L),
    "i2v_grad_post_f32": ([_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _P, _P], _I),
    "i2v_head_pool_f32": ([_P, _L, _I, _I, _I, _I, _I, _I, _P, _P], _I),
    "i2v_head_logits_ce_f32": ([_I, _I, _P, _P, _I, _P, _F, _P, _P, _P, _P], _I),
    "i2v_head_grad_f32": ([_P, _L, _I, _I, _I, _I, _I, _I, _I, _I, _P, _L, _P, _P], _I),
    "i2v_aens_coeffs_f32": ([_P, _P, _F, _I, _P], _I),
    "i2v_aens_reduce_f32": ([_P, _P, _I, _I, _P, _P, _P], _I),
}
EXPORTS = tuple(_PROTOS)


def bind(cdll):
    for name, (args, res) in _PROTOS.items():
        fn = getattr(cdll, name)
        fn.argtypes, fn.restype = args, res
    return cdll


_lib = None


def load():
    """Open the gfx950 library or fail loudly -- there is no CPU path in the product."""
    global _lib
    if _lib is None:
        # $I2V_LIB: another BUILD of the same gfx950 library (developer A/B of compile-time kernel switches); it must still answer hip:gfx950
        path = os.environ.get("I2V_LIB") or LIB_PATH
        if not os.path.exists(path):
            raise I2VError(f"{path} not found: build it with `python __graft_entry__.py` "
                           "(hipcc --offload-arch=gfx950); the attack engine has no CPU fallback")
        lib = bind(C.CDLL(path))
        if lib.i2v_backend() != HIP_BACKEND:
            raise I2VError(f"{path} reports backend {lib.i2v_backend()!r}, expected {HIP_BACKEND!r}")
        _lib = lib
    return _lib


def check(capi, status):
    if status != 0:
        raise I2VError(capi.i2v_last_error().decode())
