"""ctypes binding of libivf_hip.so (include/ivf_hip.h).

The HIP library IS the product: if it is missing or a call fails this module
raises -- there is no CPU or PyTorch fallback anywhere in the package.
PyTorch is used only for device memory and streams.
"""
import ctypes
import os
from ctypes import (POINTER, Structure, byref, c_char, c_char_p, c_float, c_int,
                    c_size_t, c_void_p)

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libivf_hip.so")


class IvfError(RuntimeError):
    pass


MATH_FP32, MATH_BF16X3, MATH_BF16X6, MATH_BF16ACT = 0, 1, 2, 3
MATH_MODES = {"fp32": MATH_FP32, "bf16x3": MATH_BF16X3, "bf16x6": MATH_BF16X6, "bf16act": MATH_BF16ACT}


class ConvDesc(Structure):
    _fields_ = [(n, c_int) for n in (
        "B", "Ti", "Hi", "Wi", "Cin", "in_ld", "in_coff", "To", "Ho", "Wo", "Cout", "out_ld",
        "out_coff", "kT", "kH", "kW", "sT", "sH", "sW", "pT", "pH", "pW", "relu", "accumulate",
        "mask_ld", "mask_coff", "d2s", "dT", "dH", "dW", "dC", "bsT", "bsH", "bsW", "math", "variant",
        "K0", "in2_ld", "in2_coff")] + [("in2", c_void_p)] + [(n, c_int) for n in ("N0", "out2_ld", "out2_coff")] + \
        [("out2", c_void_p), ("gate_out", c_void_p), ("gate_out2", c_void_p), ("gate_in", c_void_p)] + \
        [(n, c_int) for n in ("gate_out_ld", "gate_out_coff", "gate_out2_ld", "gate_in_ld", "gate_in_coff")]


class BwdGeom(Structure):
    _fields_ = [(n, c_int) for n in ("d2s", "kT", "kH", "kW", "pT", "pH", "pW", "rows")]


class PoolDesc(Structure):
    _fields_ = [(n, c_int) for n in (
        "B", "Ti", "Hi", "Wi", "C", "in_ld", "in_coff", "To", "Ho", "Wo", "out_ld", "out_coff",
        "kT", "kH", "kW", "sT", "sH", "sW", "pT", "pH", "pW", "gate_nonpos", "act_bf16")]


class I3DConfig(Structure):
    _fields_ = [(n, c_int) for n in (
        "B", "C", "T", "H", "W", "num_classes", "stem_stride_t", "pool4a_stride_t",
        "pool5a_stride_t", "head_kt", "head_kh", "head_kw", "softmax", "math")]


class TFCLSTMConfig(Structure):
    _fields_ = [(n, c_int) for n in ("B", "C", "T", "H", "W", "layers")] + [("units", c_int * 8)] + \
        [(n, c_int) for n in ("kh", "kw", "stride", "padding", "recurrent_hard_sigmoid", "only_last", "num_classes")]


class CLSTMConfig(Structure):
    _fields_ = [(n, c_int) for n in (
        "B", "C", "T", "H", "W", "hidden", "layers", "kernel", "stride", "num_classes",
        "softmax", "batch_norm", "out_step", "n_out_steps")] + [("out_steps", c_int * 16)]


_P = c_void_p
_I = c_int
_F = c_float

_SIGS = {
    "ivf_version": (c_int, []),
    "ivf_freeze_fwd": (c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "ivf_freeze_bwd_workspace_bytes": (c_size_t, [_I, _I]),
    "ivf_freeze_bwd": (c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "ivf_submask_pairs": (c_int, [_P, _I, _F, _P, _P, _P, _P]),
    "ivf_reverse_fwd": (c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "ivf_submask_pairs_batched": (c_int, [_P, _I, _I, _F, _P, _P, _P]),
    "ivf_reverse_fwd_batched": (c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "ivf_reverse_bwd": (c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "ivf_tv_norm": (c_int, [_P, _I, _I, _F, _F, _P, _P, _P]),
    "ivf_mask_reg": (c_int, [_P, _I, _I, _F, _F, _P, _P, _P, _P]),
    "ivf_adam_step": (c_int, [_P, _P, _P, _P, _I, _I, _F, _F, _F, _F, _P]),
    "ivf_search_step": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _F, _F, _F, _P]),
    "ivf_sigmoid": (c_int, [_P, _P, _I, _P]),
    "ivf_rank_frames": (c_int, [_P, _I, _I, _P, _P]),
    "ivf_init_central_select": (c_int, [_P, _P, _P, _I, _I, _I, c_float, _P, _P, _P, _P]),
    "ivf_clip_ingest_u8": (c_int, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "ivf_conv3d": (c_int, [POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P]),
    "ivf_bn_fold": (c_int, [_P, _P, _P, _P, _F, _P, _P, _I, _P]),
    "ivf_conv3d_pack_fwd_elems": (c_size_t, [_I] * 6),
    "ivf_conv3d_pack_fwd": (c_int, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "ivf_conv3d_pack_fwd_rows": (c_int, [_P, _P] + [_I] * 9 + [_P]),
    "ivf_conv3d_pack_bwd_elems": (c_size_t, [_I] * 12),
    "ivf_conv3d_pack_bwd": (c_int, [_P, _P, _P] + [_I] * 13 + [POINTER(BwdGeom), _P]),
    "ivf_conv3d_pack_bwd_fused1x1_elems": (c_size_t, [_I, _I, _I]),
    "ivf_conv3d_pack_bwd_fused1x1": (c_int, [_P, _P, _P] + [_I] * 6 + [_P]),
    "ivf_maxpool3d_fwd": (c_int, [POINTER(PoolDesc), _P, _P, _P, _P]),
    "ivf_maxpool3d_bwd": (c_int, [POINTER(PoolDesc), _P, _P, _P, _P, _I, _P]),
    "ivf_head_fwd": (c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "ivf_head_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "ivf_head_fwd_bf16": (c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "ivf_head_bwd_bf16": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "ivf_gradcam_reduce": (c_int, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "ivf_gradcam_reduce_bf16": (c_int, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "ivf_cam_resize_normalise": (c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "ivf_argmax": (c_int, [_P, _I, _I, _P, _P]),
    "ivf_i3d_create": (c_int, [POINTER(I3DConfig), POINTER(c_void_p)]),
    "ivf_i3d_destroy": (None, [_P]),
    "ivf_i3d_set_overlap": (_I, [_P, _I]),
    "ivf_i3d_weights_bytes": (c_size_t, [_P]),
    "ivf_i3d_workspace_bytes": (c_size_t, [_P]),
    "ivf_i3d_bind": (c_int, [_P, _P, _P]),
    "ivf_i3d_num_convs": (c_int, [_P]),
    "ivf_i3d_conv_info": (c_int, [_P, _I, c_char_p] + [POINTER(c_int)] * 6),
    "ivf_i3d_load_conv": (c_int, [_P, _I, _P, _P, _P, _P, _P, _P, _F, _P]),
    "ivf_i3d_forward": (c_int, [_P, _P, _I, _P, _P, _P]),
    "ivf_i3d_forward_staged": (c_int, [_P, _I, _P, _P, _P]),
    "ivf_i3d_input_buffer": (c_void_p, [_P]),
    "ivf_i3d_input_grad_buffer": (c_void_p, [_P]),
    "ivf_i3d_backward": (c_int, [_P, _I, _P, _P, _P, _P, _P]),
    "ivf_i3d_endpoint": (c_int, [_P, c_char_p, POINTER(c_void_p)] + [POINTER(c_int)] * 5),
    "ivf_i3d_act_elem_bytes": (c_int, [_P]),
    "ivf_i3d_search": (c_int, [_P, _P, _I, _P, _P, _P, _P, _F, _F, _F, _F, _F, _F, _I, _I, _I, _P, _P]),
    "ivf_i3d_perturbed_forward": (c_int, [_P, _P, _I, _P, _I, _P, _P]),
    "ivf_i3d_gradcam": (c_int, [_P, _P, _I, _P, _I, _I, _I, _P, _P, _P]),
    "ivf_i3d_gradcam_layer": (c_int, [_P, _P, _I, _P, c_char_p, _I, _I, _I, _P, _P, _P]),
    "ivf_i3d_conv_flops_per_clip": (ctypes.c_double, [_P]),
    "ivf_conv3d_variants": (c_int, [POINTER(ConvDesc), POINTER(c_int), _I]),
    "ivf_i3d_num_conv_ops": (c_int, [_P]),
    "ivf_i3d_autotune": (c_int, [_P, _I, _I, _P]),
    "ivf_i3d_get_tuning": (c_int, [_P, POINTER(c_int)]),
    "ivf_i3d_set_tuning": (c_int, [_P, POINTER(c_int)]),
    "ivf_viz_blend": (c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "ivf_viz_dots": (c_int, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "ivf_profile_enable": (c_int, [_I, _I]),
    "ivf_profile_disable": (c_int, []),
    "ivf_profile_collect": (c_int, [POINTER(ctypes.c_double), POINTER(ctypes.c_longlong),
                                    POINTER(ctypes.c_double)]),
    "ivf_profile_class_name": (c_char_p, [_I]),
    "ivf_profile_collect_sites": (c_int, [POINTER(ctypes.c_double), POINTER(ctypes.c_longlong),
                                          POINTER(ctypes.c_double), POINTER(c_int), _I]),
    "ivf_i3d_num_sites": (c_int, [_P]),
    "ivf_i3d_site_name": (c_int, [_P, _I, c_char_p]),
}

# csrc/convlstm.hip
_SIGS_OPT = {
    "ivf_clstm_create": (c_int, [POINTER(CLSTMConfig), POINTER(c_void_p)]),
    "ivf_clstm_destroy": (None, [_P]),
    "ivf_clstm_weights_bytes": (c_size_t, [_P]),
    "ivf_clstm_workspace_bytes": (c_size_t, [_P]),
    "ivf_clstm_bind": (c_int, [_P, _P, _P]),
    "ivf_clstm_load_cell": (c_int, [_P, _I] + [_P] * 12 + [_P]),
    "ivf_clstm_load_head": (c_int, [_P, _P, _P, _P, _P, _P, _P, _F, _P]),
    "ivf_clstm_forward": (c_int, [_P, _P, _I, _P, _P, _P]),
    "ivf_clstm_backward": (c_int, [_P, _I, _P, _P, _P, _P, _P]),
    "ivf_clstm_search": (c_int, [_P, _P, _I, _P, _P, _P, _P, _F, _F, _F, _F, _F, _F, _I, _I, _I, _P, _P]),
    "ivf_clstm_perturbed_forward": (c_int, [_P, _P, _I, _P, _I, _P, _P]),
    # csrc/tf_clstm.hip (documented extension)
    "ivf_tfclstm_create": (c_int, [POINTER(TFCLSTMConfig), POINTER(c_void_p)]),
    "ivf_tfclstm_destroy": (None, [_P]),
    "ivf_tfclstm_weights_bytes": (c_size_t, [_P]),
    "ivf_tfclstm_workspace_bytes": (c_size_t, [_P]),
    "ivf_tfclstm_bind": (c_int, [_P, _P, _P]),
    "ivf_tfclstm_layer_dims": (c_int, [_P, _I] + [POINTER(c_int)] * 5),
    "ivf_tfclstm_fc_inputs": (c_int, [_P]),
    "ivf_tfclstm_load_layer": (c_int, [_P, _I, _P, _P, _P, _P]),
    "ivf_tfclstm_load_head": (c_int, [_P, _P, _P, _P]),
    "ivf_tfclstm_forward": (c_int, [_P, _P, _I, _P, _P, _P]),
    "ivf_tfclstm_backward": (c_int, [_P, _I, _P, _P, _P, _P]),
    "ivf_tfclstm_perturbed_forward": (c_int, [_P, _P, _I, _P, _P, _P]),
    "ivf_tfclstm_search": (c_int, [_P, _P, _I, _P, _P, _P, _P, _F, _F, _F, _F, _F, _F, _I, _I, _P, _P]),
    "ivf_tfclstm_gradcam": (c_int, [_P, _P, _I, _P, _P, _I, _I, _I, _P, _P, _P]),
}

_lib = None


def lib():
    """Load libivf_hip.so once; raise loudly if it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise IvfError(
                f"{LIB_PATH} not found: build it with `python __graft_entry__.py` "
                "(or `make -C interpreting-video-features_amd/csrc`). There is no fallback path.")
        L = ctypes.CDLL(LIB_PATH)
        L.ivf_last_error.restype = c_char_p
        L.ivf_last_error.argtypes = []
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        for name, (res, args) in _SIGS_OPT.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def exported_symbols():
    return ["ivf_last_error"] + list(_SIGS) + list(_SIGS_OPT)


def check(rc):
    if rc != 0:
        raise IvfError(f"libivf_hip error {rc}: {lib().ivf_last_error().decode()}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return c_void_p(t.data_ptr())


def stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def require_gpu(*tensors):
    if not torch.cuda.is_available():
        raise IvfError("no HIP device visible: the saliency path runs on the MI355X only "
                       "(there is no CPU fallback)")
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise IvfError("tensor must live on the GPU (call .cuda() first)")


def f32c(t):
    """contiguous fp32 view/copy on the same device"""
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()
