"""ctypes binding of libfea_unet.so (the C ABI declared in include/fea_unet.h).

The library is built in-tree by `fea_amd/csrc/Makefile` (see
`__graft_entry__.build`).  There is NO fallback: if the shared object is missing
or a call fails, an exception is raised -- the product path never computes on
the CPU.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_longlong, c_size_t, c_void_p

import torch

from . import trace

F32, BF16 = 0, 1
_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FEA_AMD_LIB") or os.path.join(_HERE, "libfea_unet.so")   # env override: A/B builds in scripts/

P, I, L, F, Z, D = c_void_p, c_int, c_longlong, c_float, c_size_t, c_double

# name -> (restype, argtypes); mirrors include/fea_unet.h one to one.
SIGNATURES = {
    "fea_version": (I, []),
    "fea_last_error": (c_char_p, []),
    "fea_set_persistent_split": (I, [I]),
    "fea_get_persistent_split": (I, []),
    "fea_im2col3x3_nchw": (I, [P, P, I, I, I, I, I, I, P]),
    "fea_nhwc_to_nchw_f32": (I, [P, I, P, I, I, I, I, I, P]),
    "fea_nchw_f32_to_nhwc": (I, [P, P, I, I, I, I, I, I, P]),
    "fea_pack_conv3x3": (I, [P, P, I, I, I, I, P]),
    "fea_pack_conv3x3_im2col": (I, [P, P, I, I, I, I, P]),
    "fea_pack_upconv2x2": (I, [P, P, I, I, I, I, P]),
    "fea_pack_job_blocks": (I, [I, I, I, I]),
    "fea_pack_batch": (I, [P, I, I, I, P]),
    "fea_pack_conv3x3_first": (I, [P, P, I, I, I, P]),
    "fea_conv3x3_first": (I, [P, P, P, P, I, P, I, I, I, I, I, I, P]),
    "fea_conv3x3_first_rows": (I, [I, I, I, I, I, I]),
    "fea_first_wgrad_ws_bytes": (Z, [I, I, I, I]),
    "fea_conv3x3_first_wgrad": (I, [P, I, P, P, P, Z, I, I, I, I, I, I, P]),
    "fea_first_wgrad_bnbwd_rows": (I, [I, I, I, I]),
    "fea_conv3x3_first_wgrad_bnbwd": (I, [P, I, P, I, P, P, P, P, P, P, P, P, P, Z, I, I, I, I, I, I, P]),
    "fea_conv3x3": (I, [P, I, P, P, P, I, P, I, I, I, I, I, I, P]),
    "fea_conv3x3_dgrad_bnred": (I, [P, I, P, P, I, P, I, P, P, P, P, P, I, I, I, I, I, I, P]),
    "fea_pack_conv3x3_frag": (I, [P, P, I, I, I, I, P]),
    "fea_conv3x3p_rows": (I, [I, I, I, I, I, I]),
    "fea_conv3x3p": (I, [P, I, P, P, P, I, P, I, I, I, I, I, I, P]),
    "fea_conv3x3p_dgrad_bnred": (I, [P, I, P, P, I, P, I, P, P, P, P, P, I, I, I, I, I, I, P]),
    "fea_conv3x3p_bias_relu": (I, [P, I, P, P, P, I, P, I, I, I, I, I, I, I, P]),
    "fea_conv3x3_bias_relu": (I, [P, I, P, P, P, I, P, I, I, I, I, I, I, I, P]),
    "fea_conv3x3_first_bias_relu": (I, [P, P, P, P, I, I, I, I, I, I, I, P]),
    "fea_conv1x1": (I, [P, I, P, P, P, I, P, I, I, I, I, I, I, P]),
    "fea_conv_num_tiles": (I, [I, I, I]),
    "fea_upconv2x2_fwd": (I, [P, I, P, P, P, I, I, I, I, I, I, I, P]),
    "fea_upconv2x2_bwd_data": (I, [P, I, P, P, I, I, I, I, I, I, I, P]),
    "fea_upconv2x2_bwd_data_bnred": (I, [P, I, P, P, I, P, I, P, P, P, P, P, I, I, I, I, I, I, P]),
    "fea_wgrad_ws_bytes": (Z, [I, I, I, I, I, I, I]),
    "fea_conv3x3_wgrad": (I, [P, I, P, I, P, P, Z, I, I, I, I, I, I, P]),
    "fea_conv1x1_im2col_wgrad": (I, [P, I, P, I, P, P, Z, I, I, I, I, I, I, I, P]),
    "fea_upconv2x2_wgrad": (I, [P, I, P, I, P, P, Z, I, I, I, I, I, I, P]),
    "fea_bn_finalize": (I, [P, I, L, P, P, P, P, F, F, P, P, P, P, I, P]),
    "fea_bn_eval_coeffs": (I, [P, P, P, P, F, P, P, I, P]),
    "fea_bn_relu_pool": (I, [P, P, P, P, I, P, I, I, I, I, I, P]),
    "fea_bn_bwd_blocks": (I, [I, I, I, I]),
    "fea_bn_relu_bwd_reduce": (I, [P, I, P, P, P, P, P, P, P, I, I, I, I, I, P]),
    "fea_bn_bwd_finalize": (I, [P, I, L, P, P, P, P, P, I, P]),
    "fea_bn_relu_bwd_apply": (I, [P, I, P, P, P, P, P, P, P, P, P, I, I, I, I, I, P]),
    "fea_reduce_partials": (I, [P, I, I, I, P, P]),
    "fea_reduce_partials_n": (I, [P, I, I, I, I, P, P]),
    "fea_channel_sum": (I, [P, I, L, I, P, P, I, P]),
    "fea_head_fwd": (I, [P, I, P, P, P, I, I, I, I, I, I, P]),
    "fea_head_bwd_blocks": (I, [I, I, I]),
    "fea_head_bwd": (I, [P, P, I, P, P, I, P, I, I, I, I, I, I, P]),
    "fea_head_bwd_finalize": (I, [P, I, P, P, I, I, P]),
    "fea_head_loss_supported": (I, [I, I, I]),
    "fea_head_loss_sums": (I, [P, I, P, P, P, P, P, P, I, I, I, I, I, I, P]),
    "fea_head_loss_bwd": (I, [P, I, P, P, P, P, P, P, I, P, F, I, I, I, I, I, I, P]),
    "fea_head_loss_bn_supported": (I, [I, I, I, I, I]),
    "fea_head_loss_sums_bn": (I, [P, I, P, P, P, P, P, P, P, P, I, I, I, I, I, I, P]),
    "fea_head_loss_bwd_bn": (I, [P, I, P, P, P, P, P, P, P, P, I, P, F, I, I, I, I, I, I, P]),
    "fea_head_loss_bwd_bnred": (I, [P, I, P, P, P, P, P, P, P, P, P, P, P, F, I, I, I, I, I, I, P]),
    "fea_head_loss_bwd_bnapply": (I, [P, I, P, P, P, P, P, P, P, P, P, P, I, P, F, I, I, I, I, I, I, P]),
    "fea_bce_dice_sums": (I, [P, P, P, P, I, I, L, P]),
    "fea_loss_blocks": (I, [I, I, L]),
    "fea_bce_dice_sums_static": (I, [P, P, P, I, I, L, P]),
    "fea_bce_dice_grad": (I, [P, P, P, P, P, F, I, I, L, P]),
    "fea_mask_iou": (I, [P, P, P, P, P, I, I, L, P]),
    "fea_binarise": (I, [P, P, L, P]),
    "fea_opt_chunk": (I, []),
    "fea_sumsq_blocks": (I, [L]),
    "fea_grad_sumsq": (I, [P, L, P, P, P]),
    "fea_adamw_clip_step": (I, [P, I, P, P, P, P, D, D, D, D, D, D, I, P]),
    "fea_resize_ksize": (I, [I, I]),
    "fea_resize_coeffs": (I, [I, I, P, P]),
    "fea_normalize_lut": (I, [P, P, I, P]),
    "fea_resize_u8": (I, [P, I, I, I, P, P, I, I, P, P, I, P, P, I, I, P, I, P]),
}

_lib = None
# When set to a list, every `call` is bracketed by torch.cuda events on the
# current stream and (name, ev_start, ev_end, args) is appended (bench.py's
# live per-kernel timing).  None = no instrumentation.
PROFILE = None


class FeaError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """Load the shared library (once).  Raises FeaError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FeaError(
            f"{LIB_PATH} not found: build it with `make -C fea_amd/csrc` (or `python -c 'import "
            "__graft_entry__ as g; g.build()'`).  fea_amd has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def dt_code(dtype: torch.dtype) -> int:
    if dtype == torch.bfloat16:
        return BF16
    if dtype == torch.float32:
        return F32
    raise FeaError(f"unsupported compute dtype {dtype}")


def ptr(t):
    """Raw device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    if isinstance(t, int):
        return t
    return t.data_ptr()


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def call(name: str, *args):
    """Call an int-returning entry point; tensors become raw pointers; a
    negative status raises FeaError carrying fea_last_error()."""
    lib = load()
    fn = getattr(lib, name)
    conv = [ptr(a) if (a is None or isinstance(a, torch.Tensor)) else a for a in args]
    if PROFILE is not None and SIGNATURES[name][1] and SIGNATURES[name][1][-1] is P and not name.endswith(("_blocks", "_tiles", "_chunk", "_ksize", "_coeffs", "_rows", "_supported")):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = fn(*conv)
        e1.record()
        PROFILE.append((name, e0, e1, conv))
    elif trace.CALLS:
        trace.push(name)
        rc = fn(*conv)
        trace.pop()
    else:
        rc = fn(*conv)
    if SIGNATURES[name][0] is I and rc < 0 and not name.endswith(("_blocks", "_tiles", "_chunk", "_ksize", "_coeffs", "_rows", "_supported")):
        raise FeaError(f"{name} failed ({rc}): {lib.fea_last_error().decode()}")
    return rc
