"""K17 (csrc/attn_scores.hip): the probabilities of the GEMM -> softmax -> GEMM attention path with the softmax inside the score
GEMM.  Reference arithmetic: F.multi_head_attention_forward behind nn.MultiheadAttention (common/TransformerBlock.py:26):
softmax(q k^T / sqrt(d) + key padding) -> dropout -> @ v, and its autograd.  Checked against (a) an f32 torch restatement on the
same bf16 inputs with the SAME dropout mask (read back from the kernel's own outputs), and (b) the two-kernel HIP path it
replaces (case_gemm + case_softmax_*), which must see identical masks and probabilities equal to the last bf16 bit or two."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _inputs(N, heads, Lq, Lk, d, seed, packed):
    g = torch.Generator().manual_seed(seed)
    E = heads * d
    if packed:  # self-attention: one [N, L, 3E] projection tensor, as in_proj produces it
        qkv = (torch.randn(N, Lq, 3 * E, generator=g) * 0.7).to(torch.bfloat16).to(DEV)
        srcs = (qkv, qkv, qkv, 0, E, 2 * E)
    else:
        q = (torch.randn(N, Lq, E, generator=g) * 0.7).to(torch.bfloat16).to(DEV)
        kv = (torch.randn(N, Lk, 2 * E, generator=g) * 0.7).to(torch.bfloat16).to(DEV)
        srcs = (q, kv, kv, 0, 0, E)
    lens = torch.randint(max(1, Lk // 3), Lk + 1, (N,), generator=g)
    lens[0] = Lk
    valid = torch.arange(Lk)[None, :] < lens[:, None]
    if N > 2:
        valid[2] = False  # a sequence without any valid key: exact zeros everywhere
    dO = (torch.randn(N, Lq, E, generator=g)).to(torch.bfloat16).to(DEV)
    return srcs, valid.to(DEV), dO


def _heads(t, off, heads, d):
    N, L, _ = t.shape
    return t[:, :, off:off + heads * d].reshape(N, L, heads, d).permute(0, 2, 1, 3).float()


def _desc(ops, N, heads, Lq, Lk, d, q_src, k_src, v_src, drop):
    return ops._attn_desc(N, heads, Lq, Lk, d, q_src, k_src, v_src, False, 1.0 / math.sqrt(d), drop)


@pytest.mark.parametrize("N,heads,Lq,Lk,d,p_drop,packed", [
    (3, 8, 384, 384, 320, 0.1, True),     # the cfg 2 block geometry
    (30, 8, 384, 384, 320, 0.1, True),    # 720 work items on 256 persistent workgroups: the DMA stream crosses item boundaries
    (3, 8, 384, 384, 320, 0.0, True),
    (4, 2, 100, 200, 128, 0.1, False),    # ragged: a partial query tile, key quarters that end early / are empty
    (3, 3, 130, 8, 128, 0.25, False),     # one key block only
    (3, 1, 5, 376, 192, 0.0, False),
])
def test_scores_kernels_follow_the_f32_restatement(N, heads, Lq, Lk, d, p_drop, packed):
    from case_rg_amd import _abi as A, ops
    (q_src, k_src, v_src, q_off, k_off, v_off), valid, dO = _inputs(N, heads, Lq, Lk, d, 7 + Lq + d, packed)
    drop = (p_drop, 1234, 77) if p_drop > 0 else None
    ad = _desc(ops, N, heads, Lq, Lk, d, q_src, k_src, v_src, drop)
    assert A.lib.case_attention_scores_supported(ad)
    P = torch.full((N, heads, Lq, Lk), 7.0, dtype=torch.bfloat16, device=DEV)
    Pd = torch.full_like(P, 7.0) if drop else None
    kv8 = valid.to(torch.uint8)
    A.call("case_attention_scores_fwd", ad, ops._ptr(q_src, q_off), ops._ptr(k_src, k_off), ops._ptr(kv8), ops._ptr(P),
           ops._ptr(Pd) if drop else None, ops._stream())
    # f32 restatement of the forward
    q, k, v = _heads(q_src, q_off, heads, d), _heads(k_src, k_off, heads, d), _heads(v_src, v_off, heads, d)
    S = (q @ k.transpose(-1, -2)) / math.sqrt(d)
    S = S.masked_fill(~valid[:, None, None, :], float("-inf"))
    P_ref = torch.softmax(S, dim=-1)
    P_ref = torch.nan_to_num(P_ref, nan=0.0)  # rows without a valid key: zeros, as the HIP softmax defines them
    assert torch.isfinite(P.float()).all()
    assert (P.float() - P_ref).abs().max().item() <= 2.0 ** -8, (P.float() - P_ref).abs().max().item()
    assert ((P.float() - P_ref).abs() <= 2.0 ** -8 * P_ref + 1e-6).all()  # one bf16 rounding of the f32 value
    assert (P[~valid[:, None, None, :].expand_as(P)] == 0).all()
    if N > 2:
        assert (P[2] == 0).all()
    if drop:
        keep = Pd != 0
        rate = 1.0 - keep[P > 0].float().mean().item()
        assert abs(rate - p_drop) < 0.02, rate
        want = (P_ref / (1.0 - p_drop))
        assert ((Pd.float() - want).abs()[keep] <= 2.0 ** -8 * want[keep] + 1e-6).all()
        # the softmax kernel of the two-kernel path draws the same mask from the same (seed, offset)
        S32 = (S * 1.0).contiguous()
        P2, Pd2 = torch.empty_like(P), torch.empty_like(P)
        sd = ops._softmax_desc(N, heads, Lq, Lk, False, A.F32, A.BF16, drop)
        A.call("case_softmax_fwd", sd, ops._ptr(S32), ops._ptr(kv8), None, ops._ptr(P2), ops._ptr(Pd2), ops._stream())
        assert ((Pd2 != 0) == keep)[P2 > 0].all()
    # backward: dS = P (g - rowsum(g P)), g = mask(dO V^T) / (1 - p), with the kernel's own P and mask
    dS = torch.full_like(P, 7.0)
    A.call("case_attention_scores_bwd", ad, ops._ptr(dO), ops._ptr(v_src, v_off), ops._ptr(P), ops._ptr(dS), ops._stream())
    dOh = dO.reshape(N, Lq, heads, d).permute(0, 2, 1, 3).float()
    g = dOh @ v.transpose(-1, -2)
    if drop:
        g = torch.where(keep, g / (1.0 - p_drop), torch.zeros_like(g))
    Pf = P.float()
    dS_ref = Pf * (g - (g * Pf).sum(-1, keepdim=True))
    err = (dS.float() - dS_ref).abs()
    scale = dS_ref.abs().amax(dim=-1, keepdim=True).clamp_min(1e-6)
    assert torch.isfinite(dS.float()).all()
    assert (err / scale).max().item() < 1.5e-2, (err / scale).max().item()  # bf16 output rounding + f32 summation order
    rel_l2 = (dS.float() - dS_ref).norm() / dS_ref.norm().clamp_min(1e-12)
    assert rel_l2.item() < 4e-3, rel_l2.item()


def test_attention_op_uses_the_scores_kernels_and_agrees_with_the_two_kernel_path():
    """ops.attention on the GEMM + softmax + GEMM path (what training runs at head_dim 320): K17 on / off give the same output
    and gradients up to bf16 rounding; on, no f32 score tensor / case_softmax_* launch is left."""
    from case_rg_amd import _abi as A, config, ops
    import case_rg_amd
    N, heads, L, d = 2, 8, 384, 320
    (qkv, _, _, q_off, k_off, v_off), valid, dO = _inputs(N, heads, L, L, d, 5, True)
    case_rg_amd.set_dropout(True)
    calls, raw = {}, A.call

    def counting(name, *a):
        calls[name] = calls.get(name, 0) + 1
        return raw(name, *a)

    res = {}
    prev_mode = ops.ATTENTION_MODE
    try:
        ops.ATTENTION_MODE = "unfused"
        for on in (True, False):
            ops.SCORES_FUSED = on
            config.manual_seed(99)
            x = qkv.clone().requires_grad_(True)
            calls.clear()
            A.call = counting
            out = ops.attention(x, x, x, q_off, k_off, v_off, heads, d, key_valid=valid, p_drop=0.1)
            out.backward(dO)
            A.call = raw
            res[on] = (out.detach().float(), x.grad.float(), dict(calls))
    finally:
        A.call = raw
        ops.SCORES_FUSED = True
        ops.ATTENTION_MODE = prev_mode
    (o1, g1, c1), (o0, g0, c0) = res[True], res[False]
    assert c1.get("case_attention_scores_fwd") == 1 and c1.get("case_attention_scores_bwd") == 1
    assert "case_softmax_fwd" not in c1 and "case_softmax_bwd" not in c1 and c1.get("case_attention_product") == 4
    assert c1.get("case_gemm", 0) == c0.get("case_gemm", 0) - 6 == 0  # two score GEMMs and the four products around them
    assert "case_attention_scores_fwd" not in c0 and c0.get("case_softmax_fwd") == 1
    assert (o1 - o0).norm() / o0.norm() < 5e-3
    assert (g1 - g0).norm() / g0.norm() < 2e-2  # the two-kernel path rounds dP to bf16, K17 keeps it in f32


@pytest.mark.parametrize("N,heads,La,Lb,transposed,alpha", [
    (2, 8, 384, 384, False, 1.0),      # O = Pd V at the cfg 2 block geometry
    (30, 8, 384, 384, False, 0.5),     # 720 work items on 256 persistent workgroups (two to three items each)
    (30, 8, 384, 384, True, 1.0),
    (2, 8, 384, 384, True, 0.25),      # dK = alpha dS^T Q
    (3, 2, 200, 256, False, 0.5),      # a partial row tile (200 rows), contraction over 256
    (3, 2, 192, 136, True, 1.0),       # transposed: 136 output rows (a partial tile), contraction over 192
    (1, 1, 128, 128, True, 1.0),
])
def test_attention_product_kernel_against_f32_matmul(N, heads, La, Lb, transposed, alpha):
    """case_attention_product (the four bmm around the probabilities at head_dim 320): c = alpha A b with A = mat or mat^T, strided
    packed operands as the attention op passes them, rows beyond M untouched."""
    from case_rg_amd import ops
    d, E = 320, heads * 320
    g = torch.Generator().manual_seed(La * 7 + Lb)
    mat = torch.rand(N, heads, La, Lb, generator=g).to(torch.bfloat16).to(DEV)
    M, Kc = (Lb, La) if transposed else (La, Lb)
    b_src = (torch.randn(N, Kc, 3 * E, generator=g) * 0.5).to(torch.bfloat16).to(DEV)   # the operand sits in a packed projection
    out = torch.full((N, M, 2 * E), 7.0, dtype=torch.bfloat16, device=DEV)
    b_off, c_off = 2 * E, E
    assert ops.AttentionFn._product(mat, b_src, b_off, out, c_off, heads, d, M, Kc, transposed, alpha)
    A32 = mat.float().transpose(-1, -2) if transposed else mat.float()
    B32 = b_src[:, :, b_off:b_off + E].reshape(N, Kc, heads, d).permute(0, 2, 1, 3).float()
    want = alpha * (A32 @ B32)                                    # [N, heads, M, d]
    got = out[:, :, c_off:c_off + E].reshape(N, M, heads, d).permute(0, 2, 1, 3).float()
    err = (got - want).abs().max().item()
    assert err <= 2.0 ** -8 * want.abs().max().item() + 1e-3, err
    assert ((got - want).norm() / want.norm()).item() < 3e-3
    assert (out[:, :, :c_off] == 7.0).all() and (out[:, :, c_off + E:] == 7.0).all()   # neighbouring columns untouched


def test_scores_and_products_are_bit_reproducible_across_launch_geometries():
    """No atomics anywhere in K17: the same (sequence, head) must give the same bits whether its work items are the only ones of a
    launch (one item per workgroup) or ride in the middle of a persistent workgroup's stream (race screen for the cross-item prefetch)."""
    from case_rg_amd import _abi as A, ops
    N, heads, L, d = 40, 8, 384, 320
    E = heads * d
    (qkv, _, _, q_off, k_off, v_off), valid, dO = _inputs(N, heads, L, L, d, 91, True)
    kv8 = valid.to(torch.uint8)
    drop = (0.1, 4321, 10)

    def run(sl):
        q = qkv[sl].contiguous()
        n = q.shape[0]
        ad = _desc(ops, n, heads, L, L, d, q, q, q, drop)
        P, Pd, dS = (torch.empty(n, heads, L, L, dtype=torch.bfloat16, device=DEV) for _ in range(3))
        O = torch.empty(n, L, E, dtype=torch.bfloat16, device=DEV)
        G = torch.zeros(n, L, 3 * E, dtype=torch.bfloat16, device=DEV)
        A.call("case_attention_scores_fwd", ad, ops._ptr(q, q_off), ops._ptr(q, k_off), ops._ptr(kv8[sl].contiguous()), ops._ptr(P), ops._ptr(Pd),
               ops._stream())
        A.call("case_attention_scores_bwd", ad, ops._ptr(dO[sl].contiguous()), ops._ptr(q, v_off), ops._ptr(P), ops._ptr(dS), ops._stream())
        assert ops.AttentionFn._product(Pd, q, v_off, O, 0, heads, d, L, L, False)
        assert ops.AttentionFn._product(dS, q, q_off, G, k_off, heads, d, L, L, True, 0.5)
        return P, Pd, dS, O, G

    # NOTE: the dropout row keys depend on the global row index, so a slice is compared with the same rows of the full launch only
    # for the first sequences (offset 0): run the full batch twice and the leading two sequences alone
    full1, full2, head2 = run(slice(0, N)), run(slice(0, N)), run(slice(0, 2))
    for a, b in zip(full1, full2):
        assert torch.equal(a, b)
    for a, b in zip(full1, head2):
        assert torch.equal(a[:2], b)
