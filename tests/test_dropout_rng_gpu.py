"""Quality screen of the counter-based dropout generators (csrc/common.h): the element-wise kernel, the GEMM-epilogue form
(one 32-bit hash per element PAIR, 16-bit uniforms) and the attention-probability form (one row key + one multiply per column
pair), which were cheapened for speed.  For each: per-row and per-column keep rates inside binomial bounds, and |correlation|
< 0.01 between adjacent columns, adjacent rows, and the same element in two consecutive dropout sites (different counter offset).
The reference draws torch's Philox masks (F.dropout, CaSE/Model.py:69,98; TransformerBlock.py:27-28): only the statistics can be
compared, not the bits."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _corr(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    a, b = a - a.mean(), b - b.mean()
    return float((a * b).sum() / (a.norm() * b.norm() + 1e-30))


def _screen(keep, keep2, p, what):
    """keep / keep2: bool [R, C] masks of two consecutive sites."""
    R, C = keep.shape
    k = keep.double()
    rate = k.mean().item()
    sd = math.sqrt(p * (1 - p))
    assert abs(rate - (1 - p)) < 5 * sd / math.sqrt(R * C), (what, "global keep rate", rate)
    # worst row / column against the EXACT binomial tail (the normal approximation is far off for 64 columns at p = 0.1):
    # the family-wise p-value of the most extreme count must not be below 1e-4
    from scipy.stats import binom
    for axis, n, m in (("row", C, R), ("column", R, C)):
        kept = k.sum(dim=1 if axis == "row" else 0)
        lo, hi = int(kept.min().item()), int(kept.max().item())
        p_lo, p_hi = binom.cdf(lo, n, 1 - p) * m, binom.sf(hi - 1, n, 1 - p) * m
        assert p_lo > 1e-4, (what, "fewest kept in a %s" % axis, lo, "of", n, "family-wise p", p_lo)
        assert p_hi > 1e-4, (what, "most kept in a %s" % axis, hi, "of", n, "family-wise p", p_hi)
    # variance of the row rates must be binomial too (a generator that correlates the columns of a row inflates it)
    disp = (k.mean(dim=1).var().item()) / (sd * sd / C)
    assert 0.8 < disp < 1.25, (what, "row-rate dispersion / binomial", disp)
    for name, c in (("adjacent columns", _corr(k[:, :-1], k[:, 1:])), ("columns 2 apart", _corr(k[:, :-2], k[:, 2:])),
                    ("adjacent rows", _corr(k[:-1], k[1:])), ("consecutive sites", _corr(k, keep2.double())),
                    ("even/odd pair partner", _corr(k[:, 0::2], k[:, 1::2]))):
        assert abs(c) < 0.01, (what, name, c)


@pytest.fixture()
def dropout_on():
    from case_rg_amd import config
    config.set_dropout(True)
    config.manual_seed(1234)
    yield
    config.set_dropout(False)


@pytest.mark.parametrize("p", [0.1, 0.5])
def test_elementwise_dropout_statistics(dropout_on, p):
    from case_rg_amd import ops
    x = torch.ones(2048, 512, device=DEV, dtype=torch.bfloat16)
    a, b = ops.dropout(x, p), ops.dropout(x, p)
    _screen(a != 0, b != 0, p, "case_dropout p=%g" % p)


@pytest.mark.parametrize("dtype,M", [(torch.float32, 2048), (torch.bfloat16, 4096)])
def test_gemm_epilogue_dropout_statistics(dropout_on, dtype, M):
    """y = dropout(x I): the mask of the GEMM epilogue (128x128 tiling in f32, 256x256 / 64x64 tiling in bf16)."""
    from case_rg_amd import ops
    w = torch.eye(512, device=DEV)
    x = torch.ones(M, 512, device=DEV, dtype=dtype)
    a = ops.linear(x, w, None, p_drop=0.1)
    b = ops.linear(x, w, None, p_drop=0.1)
    _screen(a != 0, b != 0, 0.1, "GEMM epilogue %s" % dtype)


@pytest.mark.parametrize("mode", ["fused", "unfused"])
def test_attention_probability_dropout_statistics(dropout_on, mode):
    """Uniform scores and V = identity: O[q, :] = dropout(P)[q, :] = keep / ((1 - p) Lk), the probability mask itself."""
    from case_rg_amd import ops
    N, h, Lq, Lk, d = 4, 8, 384, 64, 64
    E = h * d
    q = torch.zeros(N, Lq, E, device=DEV, dtype=torch.bfloat16)
    kv = torch.zeros(N, Lk, 2 * E, device=DEV, dtype=torch.bfloat16)
    kv[:, :, E:] = torch.eye(Lk, device=DEV, dtype=torch.bfloat16).repeat(1, h).unsqueeze(0)
    old = ops.ATTENTION_MODE
    ops.ATTENTION_MODE = mode
    try:
        masks = []
        for _ in range(2):
            o = ops.attention(q, kv, kv, 0, 0, E, h, d, p_drop=0.1)  # [N, Lq, h*d]
            masks.append((o.view(N, Lq, h, d).permute(0, 2, 1, 3).reshape(N * h * Lq, d) != 0))
    finally:
        ops.ATTENTION_MODE = old
    _screen(masks[0], masks[1], 0.1, "attention probabilities (%s)" % mode)
    # different heads of one sequence must not share masks either
    m = masks[0].view(N, h, Lq, d).double()
    assert abs(_corr(m[:, :-1], m[:, 1:])) < 0.01
