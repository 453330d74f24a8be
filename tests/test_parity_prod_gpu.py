"""Parity at PRODUCTION tile shapes (-m gpu): the fixtures ``prod_*`` (BASELINE cfg 2 geometry: H 512, head_dim 64 / 320,
Lp 384, Lq 64, T 40, V 30522) and ``cfg5_*`` (cfg 5 geometry: d_model 768, head_dim 96 / 480, Lp 512, decoder memory
S = 20 480) were captured from the reference itself (tests/golden/gen_golden.py).  Each is replayed on the MI355X in four modes:

  fp32                f32 activations, exact-f32 MFMA                                              bar 1e-3 (north star)
  bf16_auto           what bench.py times: bf16, case_gemm's cost model, fused attention where fwd+bwd are built
  bf16_large_fused    bf16, 256x256 GEMM tiling wherever eligible, fused attention forward wherever built
  bf16_small_unfused  bf16, 128x128 GEMM tiling only, GEMM + softmax + GEMM attention

so ``gemm256_kernel``, ``fa_fwd`` / ``fa_bwd_*`` and the vector softmax run inside a test whose expected values came from the
reference.  Every comparison's measured error goes to gpurun_out/parity_errors.json (committed per round under profiles/).
bf16 bars are set from those measurements (about 2x the worst observed), per kind of tensor -- not a blanket figure."""
import numpy as np
import pytest
import torch

import cases
from helpers import load_golden, record_error, scaled_error, to_np

pytestmark = pytest.mark.gpu

MODES = {
    "fp32": dict(dtype=torch.float32, tile=0, attn="auto"),
    "bf16_auto": dict(dtype=torch.bfloat16, tile=0, attn="auto"),
    "bf16_large_fused": dict(dtype=torch.bfloat16, tile=256, attn="fused"),
    "bf16_small_unfused": dict(dtype=torch.bfloat16, tile=128, attn="unfused"),
}
# bars relative to each tensor's scale: (outputs / losses, gradients)
BARS = {"fp32": (1e-3, 1e-3), "bf16_auto": (3e-2, 6e-2), "bf16_large_fused": (3e-2, 6e-2), "bf16_small_unfused": (3e-2, 6e-2)}


HEAD_DIMS = {"prod_case_train": (64, 320), "prod_masque_train": (64, 320), "cfg5_block_5h": (480,), "cfg5_block_h": (96,),
             "cfg5_dec_layer_long_memory": (96,)}


class _Mode:
    def __init__(self, name):
        self.cfg, self.calls = MODES[name], {}

    def __enter__(self):
        import case_rg_amd
        from case_rg_amd import _abi, ops
        case_rg_amd.set_compute_dtype(self.cfg["dtype"])
        case_rg_amd.set_dropout(False)
        ops.GEMM_TILE, ops.ATTENTION_MODE, ops.TILE_TRACE = self.cfg["tile"], self.cfg["attn"], []
        self._call = _abi.call

        def counting(name, *a):
            self.calls[name] = self.calls.get(name, 0) + 1
            return self._call(name, *a)

        _abi.call = counting
        return self

    def __exit__(self, *exc):
        import case_rg_amd
        from case_rg_amd import _abi, ops
        _abi.call = self._call
        self.tiles = ops.TILE_TRACE
        ops.GEMM_TILE, ops.ATTENTION_MODE, ops.TILE_TRACE = 0, "auto", None
        case_rg_amd.set_compute_dtype(torch.float32)


@pytest.mark.parametrize("mode", list(MODES))
@pytest.mark.parametrize("name", list(cases.PROD_CASES))
def test_production_shape_case_matches_reference_fixture(name, mode):
    import case_rg_amd
    with _Mode(mode) as m:
        rec = cases.CASES[name](case_rg_amd.namespace(), torch.device("cuda"))
        torch.cuda.synchronize()
    golden = load_golden(name)
    assert set(rec) == set(golden), "case %s: keys differ: %s" % (name, set(rec) ^ set(golden))
    tol_out, tol_grad = BARS[mode]
    failures = []
    for k, want in golden.items():
        got = to_np(rec[k])
        if want.dtype.kind in "biu":
            assert np.array_equal(got, want), "%s/%s: integer / bool mismatch" % (name, k)
            continue
        tol = tol_grad if k.startswith("g") else tol_out
        rel = scaled_error("%s/%s" % (name, k), got, want)
        record_error(name, mode, k, rel, tol)
        if rel > tol:
            failures.append("%s: %.2e > %.0e" % (k, rel, tol))
    assert not failures, "%s [%s]: %s" % (name, mode, "; ".join(failures))
    # the mode really exercised the kernels it is named for
    if mode == "bf16_large_fused":
        assert 256 in m.tiles, "no GEMM of %s ran on the 256x256 tiling" % name
        from case_rg_amd import _abi
        built = [d for d in HEAD_DIMS[name] if _abi.lib.case_attention_supported(d)]
        assert (m.calls.get("case_attention_fwd", 0) > 0) == bool(built), "fused attention forward: built for %s, calls %s" % (
            built, m.calls.get("case_attention_fwd", 0))
    if mode == "bf16_small_unfused":
        assert 256 not in m.tiles and m.calls.get("case_attention_fwd", 0) == 0
    if mode == "bf16_auto" and name.startswith("prod_"):
        assert 256 in m.tiles and m.calls.get("case_attention_bwd", 0) > 0, "bench-mode kernels (gemm256, fa_bwd) did not run"
