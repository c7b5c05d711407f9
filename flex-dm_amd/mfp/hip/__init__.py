"""ctypes binding of ``libmfp_hip.so`` (the C-ABI declared in ``include/mfp_hip.h``).

The library is the *only* compute path of the product: there is no CPU or eager-PyTorch
fallback.  ``load()`` raises ``MFPHipUnavailable`` when the shared object is missing, and every
op raises when its tensors are not on a HIP device.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import (POINTER, Structure, c_char_p, c_float, c_int32, c_int64, c_size_t, c_uint8,
                    c_uint16, c_uint32, c_uint64, c_void_p)

MFP_F32, MFP_BF16 = 0, 1

GEMM_BIAS, GEMM_RELU, GEMM_RESIDUAL, GEMM_DROPOUT = 1, 2, 4, 8
GEMM_ACCUM, GEMM_ROWSKIP, GEMM_RELU_BWD, GEMM_COLSUM_A, GEMM_ROWSKIP_A = 16, 32, 64, 128, 256
MAX_LOSS_KEYS = 16

LIB_NAME = "libmfp_hip.so"
LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), LIB_NAME)


class MFPHipUnavailable(RuntimeError):
    pass


class MFPHipError(RuntimeError):
    pass


class GemmArgs(Structure):
    _fields_ = [
        ("A", c_void_p), ("B", c_void_p), ("C", c_void_p), ("bias", c_void_p),
        ("residual", c_void_p), ("aux", c_void_p), ("rowcode", c_void_p), ("colsum", c_void_p),
        ("workspace", c_void_p), ("workspace_bytes", c_size_t),
        ("M", c_int32), ("N", c_int32), ("K", c_int32),
        ("lda", c_int32), ("ldb", c_int32), ("ldc", c_int32),
        ("a_kmajor", c_int32), ("b_kmajor", c_int32),
        ("in_dtype", c_int32), ("out_dtype", c_int32), ("flags", c_int32), ("splitk", c_int32),
        ("dropout_p", c_float), ("seed", c_uint64), ("offset", c_uint64), ("step_ptr", c_void_p),
    ]


class WgradJob(Structure):
    _fields_ = [
        ("A", c_void_p), ("B", c_void_p), ("C", c_void_p), ("colsum", c_void_p), ("rowcode", c_void_p),
        ("M", c_int32), ("N", c_int32), ("lda", c_int32), ("ldb", c_int32), ("ldc", c_int32), ("_pad", c_int32),
        ("n_affine", c_void_p),
    ]


class WgradPending(Structure):
    _fields_ = [("jobs", POINTER(WgradJob)), ("njobs", c_int32), ("splitk", c_int32), ("workspace", c_void_p)]


class ReduceJob(Structure):
    _fields_ = [
        ("part", c_void_p), ("out0", c_void_p), ("out1", c_void_p), ("out2", c_void_p),
        ("split1", c_int64), ("split2", c_int64), ("N", c_int64), ("pstride", c_int64), ("P", c_int32),
    ]


class LossKey(Structure):
    _fields_ = [
        ("col_off", c_int32), ("n_feat", c_int32), ("n_class", c_int32), ("is_numerical", c_int32),
        ("target", c_void_p), ("mask", c_void_p), ("cond_idx", c_void_p),
        ("cond_stride", c_int32), ("cond_bits", c_uint32),
    ]


class MaskCol(Structure):
    _fields_ = [
        ("is_numerical", c_int32), ("n_feat", c_int32), ("input_dim", c_int32), ("group", c_int32),
        ("src", c_void_p), ("cond_idx", c_void_p), ("cond_stride", c_int32), ("cond_bits", c_uint32),
        ("idx_col", c_int32), ("_pad", c_int32), ("x_out", c_void_p), ("rowcode", c_void_p),
        ("mask_out", c_void_p),
    ]


# name -> (restype, argtypes); must list every symbol include/mfp_hip.h declares
# (tests/test_abi.py cross-checks this table against the header and the .so).
SIGNATURES = {
    "mfp_last_error": (c_char_p, []),
    "mfp_version": (c_int32, []),
    "mfp_cu_count": (c_int32, []),
    "mfp_set_reserved_cus": (c_int32, [c_int32]),
    "mfp_gemm": (c_int32, [POINTER(GemmArgs), c_void_p]),
    "mfp_gemm_workspace_bytes": (c_size_t, [POINTER(GemmArgs)]),
    "mfp_gemm_kernel_family": (c_char_p, [POINTER(GemmArgs)]),
    "mfp_wgrad_group_tiles": (c_int32, [POINTER(WgradJob), c_int32]),
    "mfp_wgrad_group_splitk": (c_int32, [POINTER(WgradJob), c_int32, c_int32, c_int32]),
    "mfp_wgrad_group_workspace_bytes": (c_size_t, [POINTER(WgradJob), c_int32, c_int32]),
    "mfp_wgrad_group": (c_int32, [POINTER(WgradJob), c_int32, c_int32, c_int32, c_void_p, c_size_t, c_void_p, c_void_p]),
    "mfp_wgrad_group_partial": (c_int32, [POINTER(WgradJob), c_int32, c_int32, c_int32, c_void_p, c_size_t, c_void_p]),
    "mfp_wgrad_reduce": (c_int32, [POINTER(WgradPending), c_int32, c_void_p]),
    "mfp_quantize_mxfp8": (c_int32, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p]),
    "mfp_gemm_mxfp8": (c_int32, [c_void_p] * 5 + [c_int32] * 6 + [c_void_p]),
    "mfp_mlp_fused_fwd": (c_int32, [c_void_p] * 13 + [c_int32, c_int32, c_float, c_float, c_uint64, c_uint64,
                                                   c_void_p, c_void_p]),
    "mfp_attn_block_fwd": (c_int32, [c_void_p] * 15 + [c_int32] * 4 + [c_float, c_float, c_uint64, c_uint64, c_void_p, c_void_p]),
    "mfp_block_fwd": (c_int32, [c_void_p] * 27 + [c_int32] * 4 + [c_float, c_float, c_uint64, c_uint64, c_uint64, c_void_p, c_void_p]),
    "mfp_block_fwd_xhat": (c_int32, [c_void_p] * 27 + [c_int32] * 4 + [c_float, c_float, c_uint64, c_uint64, c_uint64, c_void_p, c_void_p]),
    "mfp_block_fwd_xhat_half": (c_int32, [c_void_p] * 27 + [c_int32] * 4 + [c_float, c_float, c_uint64, c_uint64, c_uint64, c_void_p, c_int32, c_void_p]),
    "mfp_block_infer": (c_int32, [c_void_p] * 17 + [c_int32] * 4 + [c_float, c_void_p]),
    "mfp_ln_dense_d512": (c_int32, [c_void_p] * 9 + [c_int32, c_int32, c_int32, c_float, c_void_p]),
    "mfp_ln_dense_d512_xhat": (c_int32, [c_void_p] * 9 + [c_int32, c_int32, c_int32, c_float, c_void_p]),
    "mfp_dense_relumask_d512": (c_int32, [c_void_p] * 4 + [c_int32, c_int32, c_void_p]),
    "mfp_dense_n512_res": (c_int32, [c_void_p] * 6 + [c_int32, c_int32, c_float, c_uint64, c_uint64, c_void_p, c_void_p]),
    "mfp_dense_n512": (c_int32, [c_void_p] * 3 + [c_int32, c_int32, c_void_p]),
    "mfp_dense_n512_lnb": (c_int32, [c_void_p] * 11 + [c_int32, c_int32, c_float, c_uint64, c_uint64, c_void_p, c_void_p]),
    "mfp_dense_n512_lda": (c_int32, [c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    "mfp_attn_block_bwd": (c_int32, [c_void_p] * 9 + [c_int32] * 4 + [c_void_p]),
    "mfp_attn_block_bwd_ln": (c_int32, [c_void_p] * 17 + [c_size_t] + [c_int32] * 4 + [c_float, c_uint64, c_uint64, c_void_p, c_void_p]),
    "mfp_qkv_fused_fwd": (c_int32, [c_void_p] * 9 + [c_int32, c_int32, c_float, c_void_p]),
    "mfp_dgrad_d256": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    "mfp_dgrad_qkv": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    "mfp_encoder_dense2": (c_int32, [c_void_p] * 9 + [c_int32, c_int32, c_int32, c_void_p]),
    "mfp_dgrad_rows": (c_int32, [c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_int32, c_int32, c_int32,
                                 c_void_p, c_float, c_uint64, c_uint64, c_void_p, c_void_p]),
    "mfp_mlp_fused_bwd": (c_int32, [c_void_p] * 6 + [c_int32, c_int32, c_void_p]),
    "mfp_dgrad_qkv_ln_half": (c_int32, [c_void_p] * 9 + [c_size_t, c_int32, c_int32, c_float, c_uint64, c_uint64, c_void_p, c_void_p]),
    "mfp_mlp_bwd_ln_half": (c_int32, [c_void_p] * 12 + [c_size_t, c_int32, c_int32, c_float, c_uint64, c_uint64, c_void_p, c_void_p]),
    "mfp_mlp_bwd_ln": (c_int32, [c_void_p] * 14 + [c_size_t, c_int32, c_int32, c_float, c_uint64, c_uint64, c_void_p, c_void_p]),
    "mfp_layernorm_fwd": (c_int32, [c_void_p] * 6 + [c_int32, c_int32, c_float, c_int32, c_void_p]),
    "mfp_layernorm_bwd": (c_int32, [c_void_p] * 10 + [c_size_t, c_int32, c_int32, c_int32, c_void_p, c_void_p,
                                    c_float, c_uint64, c_uint64, c_void_p, c_void_p]),
    "mfp_layernorm_bwd_res16": (c_int32, [c_void_p] * 10 + [c_size_t, c_int32, c_int32, c_int32, c_void_p, c_void_p,
                                          c_float, c_uint64, c_uint64, c_void_p, c_void_p]),
    "mfp_layernorm_bwd_xhat": (c_int32, [c_void_p] * 9 + [c_size_t, c_int32, c_int32, c_int32, c_void_p, c_void_p,
                                         c_float, c_uint64, c_uint64, c_void_p, c_void_p]),
    "mfp_layernorm_bwd_workspace_bytes": (c_size_t, [c_int32, c_int32]),
    "mfp_layernorm_bwd_partial_rows": (c_int32, [c_int32]),
    "mfp_reduce_partials": (c_int32, [c_void_p] * 4 + [c_int64, c_int64, c_int32, c_int64, c_int64, c_void_p]),
    "mfp_reduce_partials_batch": (c_int32, [POINTER(ReduceJob), c_int32, c_void_p]),
    "mfp_attention_fwd": (c_int32, [c_void_p] * 4 + [c_int32] * 5 + [c_void_p]),
    "mfp_attention_bwd": (c_int32, [c_void_p] * 6 + [c_int32] * 5 + [c_void_p]),
    "mfp_embed_pool_fwd": (c_int32, [c_void_p] * 4 + [c_int32] * 4 + [c_void_p]),
    "mfp_embed_pool_bwd": (c_int32, [c_void_p] * 5 + [c_size_t] + [c_int32] * 4 + [c_void_p]),
    "mfp_embed_pool_bwd_workspace_bytes": (c_size_t, [c_int32] * 4),
    "mfp_embed_onehot": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "mfp_row_flags": (c_int32, [c_void_p] * 3 + [c_int32] * 3 + [c_void_p]),
    "mfp_loss_fwd_bwd": (c_int32, [c_void_p, c_void_p, c_int32, POINTER(LossKey), c_int32, c_void_p,
                                   c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "mfp_loss_fwd_bwd_sorted": (c_int32, [c_void_p, c_void_p, c_int32, POINTER(LossKey), c_int32, c_void_p,
                                          c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "mfp_sort_positions": (c_int32, [POINTER(c_void_p), POINTER(c_int32), c_void_p, c_int32, POINTER(c_int32),
                                     POINTER(c_int32), c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    "mfp_adam_num_chunks": (c_int64, [POINTER(c_int32), c_int32]),
    "mfp_adam_chunk_table": (c_int32, [POINTER(c_int32), c_int32, POINTER(c_int32), POINTER(c_int64),
                                       POINTER(c_int32), POINTER(c_int32)]),
    "mfp_adam_keras": (c_int32, [c_void_p] * 9 + [c_int64, c_void_p, c_void_p, c_void_p, c_int32, c_void_p]
                       + [c_float] * 6 + [c_void_p]),
    "mfp_transpose_cast_bf16": (c_int32, [c_void_p] * 7 + [c_int32, c_int32, c_void_p]),
    "mfp_cast_f32_bf16": (c_int32, [c_void_p, c_void_p, c_int64, c_void_p]),
    "mfp_colsum_workspace_bytes": (c_size_t, [c_int32, c_int32]),
    "mfp_dropout_bwd": (c_int32, [c_void_p] * 4 + [c_size_t, c_int32, c_int32, c_float, c_uint64,
                                                   c_uint64, c_void_p, c_int32, c_void_p]),
    "mfp_dropout_bwd_res16": (c_int32, [c_void_p] * 4 + [c_size_t, c_int32, c_int32, c_float, c_uint64,
                                                         c_uint64, c_void_p, c_void_p]),
    "mfp_colsum": (c_int32, [c_void_p] * 3 + [c_size_t] + [c_int32] * 4 + [c_void_p]),
    "mfp_step_prologue": (c_int32, [POINTER(c_float), c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_uint64, c_uint64,
                                    c_void_p, c_void_p, c_int32, c_void_p]),
    "mfp_heads_loss_partials": (ctypes.c_size_t, [c_int32]),
    "mfp_heads_loss_fwd_bwd": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, POINTER(LossKey), c_int32, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_float,
                                         c_uint64, c_uint64, c_void_p, c_void_p]),
    "mfp_heads_loss_partials_half": (ctypes.c_size_t, [c_int32]),
    "mfp_heads_loss_fwd_bwd_half": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, POINTER(LossKey), c_int32, c_void_p, c_void_p,
                                              c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_float,
                                              c_uint64, c_uint64, c_void_p, c_void_p]),
    "mfp_loss_fwd_bwd_acc": (c_int32, [c_void_p, c_void_p, c_int32, POINTER(LossKey), c_int32, c_void_p, c_void_p,
                                       c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "mfp_sample_tasks": (c_int32, [POINTER(c_float), c_int32, c_void_p, c_int32, c_uint64, c_uint64, c_void_p, c_void_p]),
    "mfp_mask_tokens": (c_int32, [POINTER(MaskCol), c_int32, c_void_p, c_int32, c_void_p, c_void_p, c_int32,
                                  c_int32, c_uint64, c_uint64, c_void_p, c_int32, c_void_p]),
    "mfp_debug_tr_probe": (c_int32, [c_void_p, c_void_p, c_void_p]),
    "mfp_debug_mx_probe": (c_int32, [c_void_p] * 6),
}

_lib = None


def load(path: str = None):
    """Load the shared library (once) and attach the prototypes.  Fails loudly."""
    global _lib
    if _lib is not None:
        return _lib
    path = path or os.environ.get("MFP_HIP_LIB", LIB_PATH)
    if not os.path.exists(path):
        raise MFPHipUnavailable(
            "%s not found at %s: build it with `python __graft_entry__.py` (or `make -C "
            "flex-dm_amd/csrc`).  The MFP hot path has no CPU fallback." % (LIB_NAME, path))
    # PyTorch's HIP runtime must be in the process BEFORE this library binds to libamdhip64: loaded the other way
    # round the two end up on different runtime instances and the first launch fails with "no ROCm-capable device"
    import torch  # noqa: F401
    try:
        lib = ctypes.CDLL(path)
    except OSError as e:  # pragma: no cover
        raise MFPHipUnavailable("cannot load %s: %s" % (path, e)) from e
    for name, (restype, argtypes) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise MFPHipUnavailable("%s does not export %s" % (path, name)) from e
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().mfp_last_error()
        raise MFPHipError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))
