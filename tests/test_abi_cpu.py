"""CPU-side checks of the C-ABI boundary: the shared library loads and exports exactly the symbols that
include/case_hip.h declares; the ctypes table mirrors the header; the product refuses to run without a GPU
instead of falling back.  No compute is launched here."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "case_hip.h")


def _declared():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(?:int|int64_t|uint32_t|const char\*)\s+(case_\w+)\s*\(", text))


def test_library_exports_every_declared_symbol():
    from case_rg_amd import _abi
    lib = ctypes.CDLL(_abi.LIB_PATH)
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), "libcase_hip.so does not export %s" % n


def test_ctypes_table_matches_header():
    from case_rg_amd import _abi
    other = {"case_version", "case_last_error", "case_gemm_tile_for", "case_optim_chunk_elems", "case_abi_features", "case_get_reserved_cus",
             "case_sizeof_opt_tensor", "case_workspace_bytes", "case_attention_bwd_scratch_floats", "case_encoder_chain_packed_bytes",
             "case_gemm_dw_slab_bytes", "case_attention_decode_mqa_workspace", "case_attention_decode_mqa_splits", "case_sizeof_step_state"}
    assert set(_abi.SIGNATURES) | other == _declared()
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    for name, args in _abi.SIGNATURES.items():
        proto = re.search(r"\b%s\s*\((.*?)\);" % name, text, flags=re.S).group(1)
        assert len([a for a in proto.split(",") if a.strip()]) == len(args), "argument count of %s" % name


def test_struct_layouts_match_header_field_order():
    from case_rg_amd import _abi
    text = open(HEADER).read()
    for cname, struct in (("CaseGemmDesc", _abi.GemmDesc), ("CaseSoftmaxDesc", _abi.SoftmaxDesc), ("CaseAttnDesc", _abi.AttnDesc)):
        body = re.search(r"typedef struct \{((?:(?!typedef struct).)*?)\} %s;" % cname, text, flags=re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if decl:
                decl = re.sub(r"^const\s+", "", decl).replace("*", " ")  # `const CaseStepState* state` -> type, name
                fields += [f.strip() for f in decl.split(None, 1)[1].split(",")]
        assert fields == [f[0] for f in struct._fields_], cname


def test_version_and_error_string():
    from case_rg_amd import _abi
    header = int(re.search(r"#define CASE_ABI_VERSION (\d+)", open(HEADER).read()).group(1))
    assert _abi.lib.case_version() == header == _abi.ABI_VERSION, "library, header and binding must be one ABI generation"
    assert isinstance(_abi.lib.case_last_error(), bytes)


def test_abi_generation_features_and_workspace_queries():
    """Round 4 (ADVICE): an out-of-tree binder can tell ABI generations apart (case_version == CASE_ABI_VERSION,
    feature mask), check its struct layouts (case_sizeof_opt_tensor) and size every caller-owned scratch (case_workspace_bytes)."""
    import ctypes as C
    from case_rg_amd import _abi, optim
    feats = _abi.lib.case_abi_features()
    for bit in (_abi.FEAT_GEMM_256, _abi.FEAT_GEMM_SMALL, _abi.FEAT_ENCODER_CHAIN, _abi.FEAT_ATTN_SCORES, _abi.FEAT_ATTN_DECODE, _abi.FEAT_OPTIM,
                _abi.FEAT_ATTN_RESIDENT, _abi.FEAT_RESERVED_CUS):
        assert feats & bit
    assert _abi.lib.case_sizeof_opt_tensor() == C.sizeof(optim._Entry) == 64
    assert _abi.lib.case_sizeof_step_state() == C.sizeof(_abi.StepState) == 64  # ABI 600: the device-resident step state
    assert _abi.StepState.rng_base.offset == 0 and _abi.StepState.step_size.offset == 8 and _abi.StepState.bc2_sqrt.offset == 12
    d = _abi.AttnDesc()
    d.N, d.heads, d.Lq, d.Lk, d.head_dim = 3, 8, 40, 4096, 64
    need = _abi.i64(0)
    _abi.call("case_attention_splitkv_workspace", d, 4, need)
    assert _abi.lib.case_workspace_bytes(_abi.WS_ATTENTION_SPLITKV, C.byref(d), 4) == need.value == 4 * 3 * 8 * 40 * 66 * 4
    assert _abi.lib.case_workspace_bytes(_abi.WS_ATTENTION_BWD, C.byref(d), 0) == _abi.lib.case_attention_bwd_scratch_floats(d) * 4 == 2 * 3 * 8 * 40 * 4
    assert _abi.lib.case_workspace_bytes(_abi.WS_OPTIM_SUMSQ, None, 17) == 68
    assert _abi.lib.case_workspace_bytes(_abi.WS_ENCODER_CHAIN_PACK, None, 0) == _abi.lib.case_encoder_chain_packed_bytes()
    assert _abi.lib.case_workspace_bytes(99, None, 0) < 0 and b"unknown kind" in _abi.lib.case_last_error()
    # reserved compute units: the one mutable setting of the library
    keep = _abi.lib.case_get_reserved_cus()
    try:
        _abi.call("case_set_reserved_cus", 16)
        assert _abi.lib.case_get_reserved_cus() == 16
        with pytest.raises(RuntimeError, match="out of range"):
            _abi.call("case_set_reserved_cus", 1000)
    finally:
        _abi.call("case_set_reserved_cus", keep)


def test_argument_validation_happens_before_any_launch():
    """A null/empty problem is rejected on the host side of the ABI (no GPU needed to see the error path)."""
    from case_rg_amd import _abi
    d = _abi.GemmDesc()
    with pytest.raises(RuntimeError, match="case_gemm"):
        _abi.call("case_gemm", d, None, None, None, None, None, None, None, None)


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-only behaviour")
def test_no_cpu_fallback():
    from case_rg_amd import ops
    with pytest.raises(RuntimeError, match="GPU only"):
        ops.linear(torch.zeros(2, 4), torch.zeros(3, 4))
    from case_rg_amd.common.TransformerEncoder import TransformerEncoderLayer
    layer = TransformerEncoderLayer(32, 8, 32, activation="gelu")
    with pytest.raises(RuntimeError, match="GPU only"):
        layer(torch.zeros(5, 2, 32))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "case_rg_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(base, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), "%s imports the oracle" % f
