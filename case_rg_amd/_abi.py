"""ctypes binding of libcase_hip.so (the C ABI declared in include/case_hip.h).

There is NO fallback: if the shared library is missing or a symbol is absent the import fails
loudly, and every call that returns a negative code raises RuntimeError(case_last_error()).
"""
import ctypes as C
import os

import torch  # noqa: F401  -- first: libcase_hip.so must bind to the HIP runtime instance torch has already loaded

_HERE = os.path.dirname(os.path.abspath(__file__))
# CASE_HIP_LIB: another build of the same library (A/B measurements of kernel variants); there is still no non-HIP path
LIB_PATH = os.environ.get("CASE_HIP_LIB") or os.path.join(_HERE, "csrc", "libcase_hip.so")

ABI_VERSION = 600  # include/case_hip.h CASE_ABI_VERSION this binding was written against
F32, BF16 = 0, 1
WS_ATTENTION_SPLITKV, WS_ATTENTION_BWD, WS_OPTIM_SUMSQ, WS_ENCODER_CHAIN_PACK, WS_GEMM_DW_SLABS = 1, 2, 3, 4, 5
(FEAT_GEMM_256, FEAT_GEMM_SMALL, FEAT_ENCODER_CHAIN, FEAT_ATTN_SCORES, FEAT_ATTN_DECODE, FEAT_OPTIM, FEAT_ATTN_RESIDENT, FEAT_RESERVED_CUS,
 FEAT_GEMM_DW_SLABS, FEAT_DECODER_CHAIN, FEAT_ATTN_DECODE_MQA, FEAT_POINTER_DECODE, FEAT_POINTER_HEAD, FEAT_GEMM_LN, FEAT_STEP_STATE,
 FEAT_INTERACTION, FEAT_ATTN_DECODE_APPEND, FEAT_LINEAR_SKINNY) = (1 << i for i in range(18))
EPI_BIAS_COL, EPI_BIAS_ROW, EPI_GELU, EPI_RELU = 1, 2, 4, 8
EPI_RESIDUAL, EPI_MUL_DGELU, EPI_MUL_DRELU, EPI_ATOMIC, EPI_DROPOUT = 16, 32, 64, 128, 256

i32, i64, u64, f32, f64, ptr = C.c_int32, C.c_int64, C.c_uint64, C.c_float, C.c_double, C.c_void_p


class GemmDesc(C.Structure):
    _fields_ = [(n, i64) for n in ("M", "N", "K", "lda", "ldb", "ldc", "ld_aux", "batch1", "batch2",
                                   "sa1", "sa2", "sb1", "sb2", "sc1", "sc2", "saux1", "saux2")] + \
               [(n, i32) for n in ("a_kmajor", "b_kmajor", "in_dtype", "out_dtype", "epilogue", "split_k", "tile")] + \
               [("alpha", f32), ("drop_p", f32), ("seed", u64), ("offset", u64), ("state", ptr)]


class StepState(C.Structure):
    """CaseStepState: the per-step scalars in caller-owned DEVICE memory (ABI 600) -- see include/case_hip.h."""
    _fields_ = [("rng_base", u64), ("step_size", f32), ("bc2_sqrt", f32), ("lr", f32), ("step", i32), ("reserved", u64 * 5)]


class EncoderChainDesc(C.Structure):
    _fields_ = [("rows", i64), ("width", i32), ("variant", i32), ("eps_ln2", f32), ("eps_ln1_next", f32)]


class SoftmaxDesc(C.Structure):
    _fields_ = [("outer", i64), ("inner", i64), ("R", i64), ("C", i64), ("causal", i32), ("in_dtype", i32),
                ("out_dtype", i32), ("drop_p", f32), ("seed", u64), ("offset", u64), ("state", ptr)]


class AttnDesc(C.Structure):
    _fields_ = [(n, i64) for n in ("N", "heads", "Lq", "Lk", "head_dim", "ldq", "ldk", "ldv", "sq", "sk", "sv", "ldo", "so")] + \
               [("causal", i32), ("scale", f32), ("drop_p", f32), ("seed", u64), ("offset", u64), ("state", ptr)]


class InteractionDesc(C.Structure):
    _fields_ = [(n, i64) for n in ("n", "Lp", "Lq", "H", "eq_div")] + [("dtype", i32)]


class AttnProductDesc(C.Structure):
    _fields_ = [(n, i64) for n in ("N", "heads", "M", "Kc", "head_dim", "lda", "sa_seq", "sa_head", "ldb", "sb_seq", "sb_head",
                                   "ldc", "sc_seq", "sc_head")] + [("a_transposed", i32), ("alpha", f32)]


# name -> argument types (all return int); mirrors include/case_hip.h one to one
SIGNATURES = {
    "case_gemm": [C.POINTER(GemmDesc), ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr],
    "case_gemm_dw_bias": [C.POINTER(GemmDesc), ptr, ptr, ptr, ptr, ptr],
    "case_gemm_dw_slabs": [C.POINTER(GemmDesc), ptr, ptr, ptr, ptr, ptr, i64, ptr],
    "case_embed_pos_fwd": [ptr, ptr, ptr, ptr, i64, i64, i64, i64, f32, f32, u64, u64, ptr, i32, ptr],
    "case_embed_pos_bwd": [ptr, ptr, ptr, i64, i64, i64, f32, f32, u64, u64, ptr, i32, ptr],
    "case_scale_add_rows": [ptr, ptr, ptr, i64, i64, i64, f32, i32, ptr],
    "case_layernorm_fwd": [ptr, ptr, ptr, ptr, ptr, ptr, ptr, i64, i64, f32, i32, ptr],
    "case_layernorm_bwd": [ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, i64, i64, i32, ptr],
    "case_layernorm_bwd_dropout": [ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, i64, i64, f32, u64, u64, ptr, i32, ptr],
    "case_layernorm_bwd_concat5": [ptr] * 15 + [i64, i64, i32, ptr],
    "case_softmax_fwd": [C.POINTER(SoftmaxDesc), ptr, ptr, ptr, ptr, ptr, ptr],
    "case_softmax_bwd": [C.POINTER(SoftmaxDesc), ptr, ptr, ptr, ptr],
    "case_attention_supported": [i64],
    "case_attention_fwd": [C.POINTER(AttnDesc), ptr, ptr, ptr, ptr, ptr, ptr, ptr],
    "case_attention_splitkv_workspace": [C.POINTER(AttnDesc), i32, C.POINTER(i64)],
    "case_attention_fwd_splitkv": [C.POINTER(AttnDesc), ptr, ptr, ptr, ptr, ptr, ptr, ptr, i64, i32, ptr],
    "case_attention_bwd_supported": [i64],
    "case_attention_decode_supported": [i64],
    "case_attention_decode": [C.POINTER(AttnDesc), ptr, ptr, ptr, ptr, ptr, ptr],
    "case_attention_decode_append": [C.POINTER(AttnDesc), ptr, ptr, ptr, ptr, ptr, i64, i64, ptr, ptr, ptr],
    "case_attention_bwd": [C.POINTER(AttnDesc), ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr],
    "case_attention_scores_supported": [C.POINTER(AttnDesc)],
    "case_attention_scores_fwd": [C.POINTER(AttnDesc), ptr, ptr, ptr, ptr, ptr, ptr],
    "case_attention_scores_bwd": [C.POINTER(AttnDesc), ptr, ptr, ptr, ptr, ptr],
    "case_attention_product_supported": [C.POINTER(AttnProductDesc)],
    "case_attention_product": [C.POINTER(AttnProductDesc), ptr, ptr, ptr, ptr],
    "case_add": [ptr, ptr, ptr, i64, i32, ptr],
    "case_add_n": [C.POINTER(ptr), i32, ptr, i64, i32, ptr],
    "case_dropout": [ptr, ptr, i64, f32, u64, u64, ptr, i32, ptr],
    "case_mask_rows": [ptr, ptr, ptr, i64, i64, i32, ptr],
    "case_colsum": [ptr, ptr, i64, i64, i32, ptr],
    "case_cast": [ptr, ptr, i64, i32, i32, ptr],
    "case_scale_cols": [ptr, ptr, ptr, i64, i64, i32, ptr],
    "case_scale_cols_bwd": [ptr, ptr, ptr, ptr, ptr, i64, i64, i32, ptr],
    "case_rowdot_fwd": [ptr, ptr, ptr, ptr, i64, i64, i32, ptr],
    "case_linear_skinny": [ptr, ptr, i32, ptr, ptr, ptr, i64, i32, i32, ptr],
    "case_rowdot_bwd": [ptr, ptr, ptr, ptr, ptr, ptr, i64, i64, i32, ptr],
    "case_masked_mean_fwd": [ptr, ptr, ptr, i64, i64, i64, i32, ptr],
    "case_masked_mean_bwd": [ptr, ptr, ptr, i64, i64, i64, i32, ptr],
    "case_highway_gate_fwd": [ptr, ptr, i64, i64, i32, ptr],
    "case_highway_gate_bwd": [ptr, ptr, ptr, i64, i64, i32, ptr],
    "case_concat5_fwd": [ptr, ptr, ptr, ptr, ptr, i64, i64, i32, ptr],
    "case_concat5_bwd": [ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, i64, i64, i32, ptr],
    "case_max_over_p_fwd": [ptr, ptr, ptr, i64, i64, i64, i32, ptr],
    "case_max_over_p_bwd": [ptr, ptr, ptr, i64, i64, i64, i32, ptr],
    "case_additive_scores_fwd": [ptr, ptr, ptr, ptr, i64, i64, i64, i64, i32, ptr],
    "case_additive_scores_bwd": [ptr, ptr, ptr, ptr, ptr, ptr, ptr, i64, i64, i64, i64, i32, ptr],
    "case_copy_scatter_fwd": [ptr, ptr, ptr, i64, i64, i64, i64, ptr],
    "case_copy_scatter_bwd": [ptr, ptr, ptr, i64, i64, i64, i64, ptr],
    "case_source_sort": [ptr, ptr, i64, i64, i64, ptr],
    "case_copy_scatter_sorted_fwd": [ptr, ptr, ptr, i64, i64, i64, i64, ptr],
    "case_nll_gather_fwd": [ptr, ptr, ptr, i64, i64, ptr],
    "case_nll_gather_bwd": [ptr, ptr, ptr, ptr, i64, i64, ptr],
    "case_row_argmax": [ptr, ptr, ptr, i64, i64, i64, ptr],
    "case_sentence_compact": [ptr, ptr, ptr, i64, i64, i64, i64, i64, ptr],
    "case_encoder_chain_pack": [ptr, ptr, ptr, ptr, ptr, ptr],
    "case_encoder_chain": [C.POINTER(EncoderChainDesc)] + [ptr] * 14,
    "case_set_reserved_cus": [i32],
    "case_attention_decode_mqa": [ptr, ptr, ptr, ptr, i64, i64, i64, i32, ptr, i64, ptr],
    "case_additive_key_exp": [ptr, ptr, i64, ptr],
    "case_pointer_attend_decode": [ptr] * 11 + [i64, i64, i64, ptr],
    "case_pointer_head_decode": [ptr] * 5 + [i32] + [ptr] * 4 + [i64, i64, i64, ptr],
    "case_gemm_ln": [C.POINTER(GemmDesc), ptr, ptr, ptr, f32, ptr, ptr, ptr, ptr, ptr, ptr],
    "case_optim_sumsq": [ptr, ptr, i64, ptr, ptr, ptr],
    "case_optim_adam_ema": [ptr, ptr, i64, ptr, f32, f64, f64, f64, f64, ptr, ptr],
    "case_step_advance": [ptr, u64, f64, f64, ptr],
    "case_interaction_supported": [C.POINTER(InteractionDesc)],
    "case_interaction_fwd": [C.POINTER(InteractionDesc), ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr],
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "case_rg_amd: %s is missing. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C case_rg_amd/csrc`). There is no CPU or eager fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    lib.case_version.restype = C.c_int
    if lib.case_version() != ABI_VERSION:
        # a layout-only change (CaseOptTensor grew in round 3) would otherwise mis-stride tables silently
        raise ImportError("case_rg_amd: %s is ABI generation %d, this package binds %d -- rebuild it (make -C case_rg_amd/csrc)"
                          % (LIB_PATH, lib.case_version(), ABI_VERSION))
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.argtypes = args
        fn.restype = C.c_int
    lib.case_optim_chunk_elems.restype = C.c_int
    lib.case_optim_chunk_elems.argtypes = []
    lib.case_encoder_chain_packed_bytes.restype = C.c_int64
    lib.case_encoder_chain_packed_bytes.argtypes = []
    lib.case_attention_bwd_scratch_floats.restype = C.c_int64
    lib.case_attention_bwd_scratch_floats.argtypes = [C.POINTER(AttnDesc)]
    lib.case_attention_decode_mqa_workspace.restype = C.c_int64
    lib.case_attention_decode_mqa_workspace.argtypes = [i64, i64, i32]
    lib.case_attention_decode_mqa_splits.restype = C.c_int32
    lib.case_attention_decode_mqa_splits.argtypes = [i64, i64]
    lib.case_abi_features.restype = C.c_uint32
    lib.case_abi_features.argtypes = []
    lib.case_get_reserved_cus.restype = C.c_int
    lib.case_get_reserved_cus.argtypes = []
    lib.case_sizeof_opt_tensor.restype = C.c_int
    lib.case_sizeof_opt_tensor.argtypes = []
    lib.case_sizeof_step_state.restype = C.c_int
    lib.case_sizeof_step_state.argtypes = []
    if lib.case_sizeof_step_state() != C.sizeof(StepState):
        raise ImportError("case_rg_amd: CaseStepState is %d bytes in %s, %d in this binding" % (lib.case_sizeof_step_state(), LIB_PATH, C.sizeof(StepState)))
    lib.case_gemm_dw_slab_bytes.restype = C.c_int64
    lib.case_gemm_dw_slab_bytes.argtypes = [C.POINTER(GemmDesc)]
    lib.case_workspace_bytes.restype = C.c_int64
    lib.case_workspace_bytes.argtypes = [i32, ptr, i64]
    lib.case_gemm_tile_for.restype = C.c_int  # 128 / 256 or a negative code: not routed through check()
    lib.case_gemm_tile_for.argtypes = [C.POINTER(GemmDesc), ptr, ptr, ptr, ptr, ptr, ptr]
    lib.case_last_error.restype = C.c_char_p
    return lib


lib = _load()


def check(code, what):
    if code != 0:
        raise RuntimeError("%s failed (%d): %s" % (what, code, lib.case_last_error().decode()))


def call(name, *args):
    check(getattr(lib, name)(*args), name)
