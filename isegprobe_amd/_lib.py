"""ctypes loader for libisegprobe_hip.so (the C-ABI declared in include/isegprobe_hip.h).

There is no fallback: if the library is missing the product path raises.  Build it with
``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C isegprobe_amd/csrc``.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ISEGPROBE_HIP_LIB") or os.path.join(_HERE, "csrc", "libisegprobe_hip.so")  # env override: kernel A/B experiments

ABI_VERSION = 19

ISP_F32, ISP_BF16, ISP_F16 = 0, 1, 2
EP_BIAS_BF16, EP_BIAS_RELU_BF16, EP_BIAS_GELU_BF16, EP_BIAS_F32, EP_RESIDUAL_F32, EP_TOKENS_F32, EP_AXPY_RES_BF16, EP_BIAS_TAPS_RELU_BF16, EP_RELU_DOT_PARTIAL_F32, EP_BIAS_QGELU_BF16, EP_BIAS_GELU_SAVE_BF16, EP_MUL_DGELU_BF16, EP_BIAS_QGELU_SAVE_BF16, EP_MUL_DQGELU_BF16, EP_AXPY_RES_STATS_BF16, EP_LNFOLD_BF16, EP_LNFOLD_GELU_BF16, EP_RESIDUAL_STATS_F32, EP_LNFOLD_LAYERNORM_BF16, EP_BIAS_RELU_STATS_BF16 = range(20)

_ERR = {-1: "invalid argument", -2: "unsupported configuration", -3: "HIP launch failed"}


class IspError(RuntimeError):
    pass


class Epilogue(ctypes.Structure):
    _fields_ = [
        ("kind", ctypes.c_int),
        ("out", ctypes.c_void_p),
        ("ldo", ctypes.c_long),
        ("bias", ctypes.c_void_p),
        ("gamma", ctypes.c_void_p),
        ("pos", ctypes.c_void_p),
        ("tokens_per_image", ctypes.c_int),
        ("res", ctypes.c_void_p),
        ("alpha", ctypes.c_float),
        ("img_h", ctypes.c_int),
        ("img_w", ctypes.c_int),
        ("out2", ctypes.c_void_p),
        ("out3", ctypes.c_void_p),
        ("alpha2", ctypes.c_float),
    ]


_vp, _i, _l, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float
_EP = ctypes.POINTER(Epilogue)

# name -> argtypes; must list every symbol include/isegprobe_hip.h declares
SIGNATURES = {
    "isp_abi_version": [],
    "isp_click_maps_fwd": [_vp, _vp, _i, _i, _i, _i, _f, _f, _i, _i, _vp],
    "isp_normalize_fwd": [_vp, _vp, _vp, _i, _i, _i, _i, ctypes.POINTER(_f), ctypes.POINTER(_f), _vp],
    "isp_patchify_fwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "isp_gemm_bf16": [_vp, _l, _vp, _l, _i, _i, _EP, _vp],
    "isp_gemm_f16": [_vp, _l, _vp, _l, _i, _i, _EP, _vp],
    "isp_conv3x3_nhwc_bf16": [_vp, _vp, _i, _i, _i, _i, _i, _EP, _vp],
    "isp_conv3x3_nhwc_f16": [_vp, _vp, _i, _i, _i, _i, _i, _EP, _vp],
    "isp_conv3x3_partial_slots": [_i],
    "isp_conv3x3_of_bilinear_supported": [_i, _i, _i, _i, _i, _i],
    "isp_conv3x3_of_bilinear_blend": [_vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "isp_conv3x3_of_bilinear_bwd_workspace_bytes": [_i, _i, _i, _i],
    "isp_conv3x3_of_bilinear_blend_bwd": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "isp_sum_partials_f32": [_vp, _vp, _l, _i, _f, _vp],
    "isp_layernorm_fwd": [_vp, _vp, _vp, _vp, _l, _i, _f, _i, _i, _i, _i, _l, _l, _vp],
    "isp_attention_fwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i] + [_l] * 9 + [_f, _vp],
    "isp_attention_fwd_f16": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i] + [_l] * 9 + [_f, _vp],
    "isp_attention_fwd_logit2_f16": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i] + [_l] * 9 + [_vp],
    "isp_attention_fwd_logit2": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i] + [_l] * 9 + [_vp],
    "isp_gemm_stats_slots": [],
    "isp_conv_stats_slots": [_i],
    "isp_gemm_f16_stats_slots": [_l, _i],
    "isp_attention_pipe_supported": [_i, _i, _i, _l],
    "isp_attention_fwd_pipe": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i] + [_l] * 9 + [_i, _vp],
    "isp_attention_fwd_lse": [_vp, _vp, _vp, _vp, _vp, _l, _i, _i, _i, _i, _i] + [_l] * 9 + [_f, _vp],
    "isp_attention_bwd": [_vp] * 7 + [_l] + [_vp] * 3 + [_i] * 5 + [_l] * 9 + [_f, _vp, _vp],
    "isp_conv3x3_wgrad_bf16_atomic": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "isp_next_points_workspace_bytes": [_i, _i, _i],
    "isp_next_points": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp],
    "isp_jbu_kernels_f32": [_vp] * 7 + [_f, _f, _i, _i, _i, _vp],
    "isp_bicubic_x2_nhwc_f32": [_vp, _vp, _i, _i, _i, _i, _vp],
    "isp_adaptive_conv7_nhwc_f32": [_vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "isp_split_bf16x3": [_vp, _l, _vp, _l, _i, _i, _i, _i, _f, _vp],
    "isp_softmax_rows_f32": [_vp, _l, _i, _l, _vp],
    "isp_attention_packed_f32": [_vp, _vp, _i, _i, _i, _f, _vp],
    "isp_probe_mfma_bf16": [_vp, _vp, _i, _i, _vp],
    "isp_probe_mfma_bf16_32x32": [_vp, _vp, _i, _i, _vp],
    "isp_probe_copy": [_vp, _vp, _l, _vp],
    "isp_robot_click_workspace_bytes": [_i, _i],
    "isp_robot_click": [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp],
    "isp_threshold_u8": [_vp, _vp, _f, _l, _vp],
    "isp_resize_bilinear_ac_nchw_f32_bwd": [_vp, _vp, _l, _i, _i, _i, _i, _vp],
    "isp_layernorm_wgrad": [_vp, _i, _l, _vp, _l, _vp, _vp, _l, _i, _f, _vp],
    "isp_layernorm_bwd": [_vp, _i, _l, _vp, _l, _vp, _vp, _l, _vp, _l, _l, _i, _f, _i, _i, _i, _vp],
    "isp_resize_bilinear_ac_nhwc_bf16": [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "isp_resize_bilinear_ac_nchw_f32": [_vp, _vp, _l, _i, _i, _i, _i, _l, _vp],
    "isp_resize_nhwc_bf16": [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "isp_token_add_fwd": [_vp, _i, _vp, _i, _l, _i, _i, _i, _vp],
    "isp_adaptive_avg_pool_nchw_f32": [_vp, _vp, _l, _i, _i, _i, _i, _vp],
    "isp_jbu_range_proj": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "isp_jbu_kernels": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _i, _i, _i, _vp, _vp],
    "isp_jbu_apply": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "isp_bf16_to_f16": [_vp, _vp, _l, _vp],
    "isp_jbu_apply_bwd": [_vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "isp_jbu_blend": [_vp, _vp, _i, _i, _i, _i, _i, _vp],
    "isp_jbu_kernels_resized": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _i, _i, _i, _i, _i, _vp, _vp],
    "isp_jbu_apply_resized": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "isp_fuse_flip_sigmoid": [_vp, _vp, _l, _i, _i, _i, _vp],
    "isp_minmax_nchw_f32": [_vp, _vp, _vp, _i, _i, _l, _vp],
    "isp_loftup_fourier_cn": [_vp] * 8 + [_i, _i, _i, _i, _i, _f, _vp],
    "isp_loftup_fourier_cn_f32": [_vp] * 8 + [_i, _i, _i, _i, _i, _f, _vp],
    "isp_loftup_fourier_cn_f16": [_vp] * 8 + [_i, _i, _i, _i, _i, _f, _vp],
    "isp_conv3x3_s2_c32": [_vp, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "isp_vit_mlp_fused": [_vp, _vp, _vp, _vp, _vp, _l, _i, _i, _f, _vp],
    "isp_vit_mlp_fused_rows": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _i, _vp],
    "isp_bn_train_stats": [_vp, _vp, _l, _i, _vp],
    "isp_bn_train_apply": [_vp, _vp, _vp, _vp, _vp, _l, _i, _f, _i, _vp, _vp, _vp, _vp, _i, _f, _vp],
    "isp_bn_train_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _l, _i, _f, _vp],
    "isp_adaptive_max_pool_nhwc_bf16": [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "isp_tn_gemm_bf16_atomic": [_vp, _l, _vp, _l, _vp, _l, _l, _i, _i, _i, _i, _i, _i, _i, _vp],
    "isp_relu_mask_colsum": [_vp, _vp, _vp, _vp, _l, _i, _vp],
    "isp_classifier_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _l, _i, _i, _vp],
    "isp_resize_bilinear_ac_nhwc_bwd": [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "isp_classifier_fwd": [_vp, _vp, _f, _vp, _l, _i, _vp],
    "isp_nhwc_bf16_to_nchw_f32": [_vp, _vp, _i, _i, _l, _vp],
    "isp_nchw_f32_to_nhwc_bf16": [_vp, _vp, _i, _i, _l, _l, _l, _l, _vp],
}

_lib = None


LONG_RETURNS = {"isp_robot_click_workspace_bytes", "isp_next_points_workspace_bytes", "isp_conv3x3_of_bilinear_bwd_workspace_bytes"}


def lib():
    """Load (once) and return the bound library; raises if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise IspError(
                f"{LIB_PATH} is missing: the HIP extension is not built and there is no CPU fallback. "
                "Run `python -c 'import __graft_entry__ as g; g.build()'`."
            )
        handle = ctypes.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the symbol is not exported
            fn.argtypes = argtypes
            fn.restype = ctypes.c_long if name in LONG_RETURNS else ctypes.c_int
        if handle.isp_abi_version() != ABI_VERSION:
            raise IspError(f"ABI mismatch: library {handle.isp_abi_version()} != binding {ABI_VERSION}")
        _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        raise IspError(f"{what} failed: {_ERR.get(rc, rc)}")
