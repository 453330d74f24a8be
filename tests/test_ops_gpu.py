"""Kernel-level numerics on the MI355X: every C-ABI op (through its autograd shell in case_rg_amd.ops)
against a plain PyTorch fp32 reference of the same op, forward and gradients.

Tolerances: f32 mode 1e-3 relative to the tensor scale (north-star bar; the exact-f32 MFMA path is usually
~1e-6); bf16 mode is reported against its own bar (3e-2 of the tensor scale, bf16 has 8 mantissa bits)."""
import ctypes
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _ops():
    from case_rg_amd import config, ops
    config.set_dropout(False)
    return ops


def _close(got, want, tol, what):
    got, want = got.float(), want.float()
    scale = want.abs().max().item() + 1e-6
    err = (got - want).abs().max().item()
    assert err <= tol * scale, "%s: max err %.3e vs scale %.3e (tol %.1e)" % (what, err, scale, tol)


def _tol(dt):
    return 1e-3 if dt == torch.float32 else 3e-2


def _rand(*shape, dt=torch.float32, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV).to(dt)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,K,N", [(256, 512, 384), (300, 136, 200), (7, 20, 5), (129, 64, 1), (1024, 2560, 512)])
def test_linear_fwd_bwd(dt, M, K, N):
    ops = _ops()
    x = _rand(M, K, dt=dt, seed=1).requires_grad_()
    w = _rand(N, K, seed=2, scale=K ** -0.5).requires_grad_()
    b = _rand(N, seed=3).requires_grad_()
    res = _rand(M, N, dt=dt, seed=4).requires_grad_()
    y = ops.linear(x, w, b, residual=res)
    xr, wr, br, rr = [t.detach().float().requires_grad_() for t in (x, w.to(dt), b, res)]
    yr = F.linear(xr, wr, br) + rr
    _close(y, yr, _tol(dt), "linear y")
    g = _rand(M, N, dt=dt, seed=5)
    y.backward(g)
    yr.backward(g.float())
    _close(x.grad, xr.grad, _tol(dt), "linear dx")
    _close(w.grad, wr.grad, _tol(dt), "linear dw")
    _close(b.grad, br.grad, _tol(dt), "linear db")
    _close(res.grad, rr.grad, _tol(dt), "linear dres")


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("act", ["gelu", "relu"])
def test_ffn(dt, act):
    ops = _ops()
    M, K, Fh, N = 200, 96, 72, 96
    x = _rand(M, K, dt=dt, seed=1).requires_grad_()
    w1, b1 = _rand(Fh, K, seed=2, scale=K ** -0.5).requires_grad_(), _rand(Fh, seed=3).requires_grad_()
    w2, b2 = _rand(N, Fh, seed=4, scale=Fh ** -0.5).requires_grad_(), _rand(N, seed=5).requires_grad_()
    y = ops.ffn(x, w1, b1, w2, b2, act, residual=x)
    ps = [x, w1, b1, w2, b2]
    rs = [t.detach().to(dt).float().requires_grad_() if t.dim() == 2 and t is not x else t.detach().float().requires_grad_() for t in ps]
    fa = F.gelu if act == "gelu" else F.relu
    yr = F.linear(fa(F.linear(rs[0], rs[1], rs[2])), rs[3], rs[4]) + rs[0]
    _close(y, yr, _tol(dt), "ffn y")
    g = _rand(M, N, dt=dt, seed=6)
    y.backward(g)
    yr.backward(g.float())
    for name, a, r in zip(["dx", "dw1", "db1", "dw2", "db2"], ps, rs):
        _close(a.grad, r.grad, 2 * _tol(dt), "ffn " + name)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("R,C", [(33, 32), (100, 2560), (5, 160), (101, 256), (77, 768), (39, 1280), (20, 3840)])  # 256 / 768 / 1280: the 8-byte-vector backward
def test_layernorm(dt, R, C):
    ops = _ops()
    x = _rand(R, C, dt=dt, seed=1).requires_grad_()
    x2 = _rand(R, C, dt=dt, seed=2).requires_grad_()
    g, b = (1 + 0.1 * _rand(C, seed=3)).requires_grad_(), _rand(C, seed=4).requires_grad_()
    y = ops.layer_norm(x, g, b, add=x2)
    xr, x2r, gr, br = [t.detach().float().requires_grad_() for t in (x, x2, g, b)]
    yr = F.layer_norm(xr + x2r, (C,), gr, br)
    _close(y, yr, _tol(dt), "ln y")
    go = _rand(R, C, dt=dt, seed=5)
    y.backward(go)
    yr.backward(go.float())
    _close(x.grad, xr.grad, _tol(dt), "ln dx")
    _close(x2.grad, x2r.grad, _tol(dt), "ln dx2")
    _close(g.grad, gr.grad, _tol(dt), "ln dgamma")
    _close(b.grad, br.grad, _tol(dt), "ln dbeta")


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N,h,Lq,Lk,d,causal", [(3, 8, 7, 7, 4, False), (2, 8, 5, 5, 4, True), (2, 8, 40, 72, 64, False),
                                                 (2, 8, 96, 96, 320, False), (2, 4, 13, 200, 20, False)])
def test_attention(dt, N, h, Lq, Lk, d, causal):
    ops = _ops()
    E = h * d
    self_attn = Lq == Lk
    valid = torch.ones(N, Lk, dtype=torch.bool, device=DEV)
    valid[1, Lk // 2 + 1:] = False
    if self_attn:
        qkv = _rand(N, Lq, 3 * E, dt=dt, seed=1, scale=0.5).requires_grad_()
        o = ops.attention(qkv, qkv, qkv, 0, E, 2 * E, h, d, key_valid=valid, causal=causal)
        r = qkv.detach().float().requires_grad_()
        q, k, v = r.split(E, dim=-1)
    else:
        qs = _rand(N, Lq, E, dt=dt, seed=1, scale=0.5).requires_grad_()
        kv = _rand(N, Lk, 2 * E, dt=dt, seed=2, scale=0.5).requires_grad_()
        o = ops.attention(qs, kv, kv, 0, 0, E, h, d, key_valid=valid, causal=causal)
        rq, rkv = qs.detach().float().requires_grad_(), kv.detach().float().requires_grad_()
        q, (k, v) = rq, rkv.split(E, dim=-1)
    qh, kh, vh = [t.reshape(N, -1, h, d).transpose(1, 2) for t in (q, k, v)]
    s = qh @ kh.transpose(-1, -2) / math.sqrt(d)
    s = s.masked_fill(~valid[:, None, None, :], float("-inf"))
    if causal:
        s = s + torch.triu(torch.full((Lq, Lk), -1e20, device=DEV), 1)
    orf = (torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(N, Lq, E)
    _close(o, orf, _tol(dt), "attn o")
    g = _rand(N, Lq, E, dt=dt, seed=3)
    o.backward(g)
    orf.backward(g.float())
    if self_attn:
        _close(qkv.grad, r.grad, 2 * _tol(dt), "attn dqkv")
    else:
        _close(qs.grad, rq.grad, 2 * _tol(dt), "attn dq")
        _close(kv.grad, rkv.grad, 2 * _tol(dt), "attn dkv")


@pytest.mark.parametrize("N,h,Lk", [(32, 8, 3840), (40, 8, 37), (24, 8, 64), (300, 1, 129)])
def test_attention_decode_step(N, h, Lk):
    """One query per sequence against cached keys / values (the greedy step's cross- and self-attention): the streaming kernel
    against torch and against the general fused forward; ragged validity, one sequence without any valid key (exact zeros)."""
    ops = _ops()
    d, E, dt = 64, h * 64, torch.bfloat16
    q = _rand(N, 1, E, dt=dt, seed=1, scale=0.5)
    kv = _rand(N, Lk, 2 * E, dt=dt, seed=2, scale=0.5)
    g = torch.Generator().manual_seed(3)
    valid = (torch.rand(N, Lk, generator=g) > 0.2).to(DEV)
    valid[0] = True
    valid[1] = False
    valid[2, Lk // 3:] = False
    assert N * h >= ops.DECODE_MIN_PAIRS
    with torch.no_grad():
        o = ops.attention(q, kv, kv, 0, 0, E, h, d, key_valid=valid)
        keep, ops.DECODE_MIN_PAIRS = ops.DECODE_MIN_PAIRS, 1 << 30
        try:
            general = ops.attention(q, kv, kv, 0, 0, E, h, d, key_valid=valid)
        finally:
            ops.DECODE_MIN_PAIRS = keep
    k, v = kv.float().split(E, dim=-1)
    qh, kh, vh = [t.reshape(N, -1, h, d).transpose(1, 2) for t in (q.float(), k, v)]
    s = (qh @ kh.transpose(-1, -2) / math.sqrt(d)).masked_fill(~valid[:, None, None, :], float("-inf"))
    p = torch.softmax(s, -1).nan_to_num(0.0)
    ref = (p @ vh).transpose(1, 2).reshape(N, 1, E)
    _close(o, ref, _tol(dt), "decode attention vs torch")
    _close(o, general.float(), 1e-2, "decode attention vs the general fused forward")
    assert torch.count_nonzero(o[1]) == 0, "no valid key: exact zeros"


@pytest.mark.parametrize("N,h,T,t", [(32, 8, 64, 0), (32, 8, 64, 17), (32, 8, 64, 63), (256, 8, 40, 5), (200, 1, 129, 128)])
def test_attention_decode_append_equals_copy_then_attention(N, h, T, t):
    """The greedy step's self-attention with the cache append inside the launch (case_attention_decode_append) against the strided copy into the
    cache followed by case_attention_decode: the output and the WHOLE cache bit for bit -- position t written, every other row untouched --
    with ragged history masks, a PAD token at position t (written, not attended) and a sequence without any valid position (exact zeros)."""
    ops = _ops()
    d, E, dt = 64, h * 64, torch.bfloat16
    qkv = _rand(N, 1, 3 * E, dt=dt, seed=11, scale=0.5)
    cache0 = _rand(N, T, 2 * E, dt=dt, seed=12, scale=0.5)
    g = torch.Generator().manual_seed(13)
    valid = (torch.rand(N, T, generator=g) > 0.2)
    valid[:, t + 1:] = False  # positions behind t do not exist yet
    valid[:, t] = True
    valid[1] = False          # nothing to attend: exact zeros
    valid[2, t] = False       # a PAD token at position t: its K / V are cached, not attended
    valid = valid.to(DEV)
    assert ops.decode_append_supported(qkv, cache0, h, d) or torch.is_grad_enabled()
    with torch.no_grad():
        assert ops.decode_append_supported(qkv, cache0, h, d)
        a = cache0.clone()
        o1 = ops.attention_decode_append(qkv, a, t, h, d, key_valid=valid)
        b = cache0.clone()
        b[:, t] = qkv[:, 0, E:]
        o2 = ops.attention(qkv, b, b, 0, 0, E, h, d, key_valid=valid)
    assert torch.equal(a, b), "the cache after the fused append differs from the copied one"
    assert torch.equal(o1, o2), "attention output differs: max |diff| %.3e" % (o1.float() - o2.float()).abs().max().item()
    assert torch.count_nonzero(o1[1]) == 0
    with pytest.raises((ValueError, RuntimeError)):
        with torch.no_grad():
            ops.attention_decode_append(qkv, a, T, h, d, key_valid=valid)  # position outside the cache


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,widths,nout", [(256, (512, 512, 512), 3), (33, (512,), 1), (7, (96, 40, 8, 24), 8), (300, (768, 768), 2)])
def test_linear_skinny_equals_linear_on_the_concatenation(dt, rows, widths, nout):
    """case_linear_skinny (the greedy step's mixing logits) against torch's Linear on the concatenated rows in f32: 1-4 inputs, 1-8 outputs,
    widths that are not multiples of 64, more rows than waves in flight; and against the cat + GEMM path it replaces."""
    ops = _ops()
    xs = [_rand(rows, 1, w, dt=dt, seed=20 + i, scale=0.7) for i, w in enumerate(widths)]
    W = _rand(nout, sum(widths), dt=torch.float32, seed=31, scale=0.1)
    b = _rand(nout, dt=torch.float32, seed=32)
    with torch.no_grad():
        assert ops.linear_skinny_supported(xs, W)
        y = ops.linear_skinny(xs, W, b)
        old = ops.linear(torch.cat(xs, dim=-1), W, b, out_dtype=torch.float32)
    ref = torch.cat([x.float() for x in xs], dim=-1) @ W.t() + b
    assert y.shape == (rows, 1, nout) and y.dtype == torch.float32
    _close(y, ref, 1e-4, "skinny linear vs f32 torch")  # (the inputs are exact in both: only the summation order differs)
    _close(old, ref, 1e-4 if dt == torch.float32 else 2e-2, "cat + GEMM path vs f32 torch")
    assert not ops.linear_skinny_supported(xs + xs + xs, W)  # > 4 inputs: the caller concatenates


def test_softmax_masks_and_empty_rows():
    ops = _ops()
    x = _rand(2, 6, 9, seed=1).requires_grad_()
    cv = torch.ones(2, 9, dtype=torch.bool, device=DEV)
    cv[0, 5:] = False
    rv = torch.ones(2, 6, dtype=torch.bool, device=DEV)
    rv[1, 4:] = False
    p = ops.masked_softmax(x, cv, rv, outer=2)
    xr = x.detach().clone().requires_grad_()
    m = rv[:, :, None] & cv[:, None, :]
    pr = torch.softmax(xr.masked_fill(~m, float("-inf")), -1).masked_fill(~m, 0.0)
    _close(p, pr, 1e-5, "softmax p")
    assert (p[1, 4:] == 0).all(), "fully masked rows must be exactly zero"
    g = _rand(2, 6, 9, seed=2)
    p.backward(g)
    pr.backward(g)
    _close(x.grad, torch.nan_to_num(xr.grad), 1e-5, "softmax dx")
    assert (x.grad[1, 4:] == 0).all()
    # wide rows take the workgroup-per-row kernel
    xw = _rand(3, 3000, seed=3)
    _close(ops.masked_softmax(xw.view(1, 3, 3000)), torch.softmax(xw, -1).view(1, 3, 3000), 1e-5, "softmax wide")


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C", [2560, 70, 256, 768])
def test_layernorm_carry_sums_both_gradients_of_x(dt, C):
    """x + f(LN(x)) with layer_norm_carry: the residual gradient is added inside the backward kernel (vector and scalar)."""
    ops = _ops()
    R = 37
    x = _rand(R, C, dt=dt, seed=1).requires_grad_()
    g, b = (1 + 0.1 * _rand(C, seed=2)).requires_grad_(), _rand(C, seed=3).requires_grad_()
    y, xc = ops.layer_norm_carry(x, g, b)
    out = 0.5 * y + xc
    go = _rand(R, C, dt=dt, seed=4)
    out.backward(go)
    xr, gr, br = [t.detach().float().requires_grad_() for t in (x, g, b)]
    outr = 0.5 * F.layer_norm(xr, (C,), gr, br) + xr
    outr.backward(go.float())
    _close(out, outr, _tol(dt), "ln carry out")
    _close(x.grad, xr.grad, _tol(dt), "ln carry dx")
    _close(g.grad, gr.grad, 2 * _tol(dt), "ln carry dgamma")


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("H", [64, 1000, 50])
def test_masked_mean(dt, H):
    """universal_sentence_embedding: mean over valid positions; H = 64 / 1000 take the vector kernels, 50 (bf16) the scalar one."""
    ops = _ops()
    n, L = 5, 37
    x = _rand(n, L, H, dt=dt, seed=1).requires_grad_()
    valid = torch.zeros(n, L, dtype=torch.bool, device=DEV)
    for i, k in enumerate((37, 1, 20, 9, 30)):
        valid[i, :k] = True
    y = ops.masked_mean(x, valid)
    xr = x.detach().float().requires_grad_()
    yr = (xr * valid[:, :, None]).sum(1) / valid.sum(1, keepdim=True)
    _close(y, yr, _tol(dt), "masked mean")
    g = _rand(n, H, dt=dt, seed=2)
    y.backward(g)
    yr.backward(g.float())
    _close(x.grad, xr.grad, _tol(dt), "masked mean dx")
    assert (x.grad[1, 1:] == 0).all()


@pytest.mark.parametrize("C", [40, 384, 640, 1024])
def test_softmax_bf16_row_in_registers(C):
    """bf16 rows with C % 8 == 0 and C <= 1024 take the wave-per-row vector kernels: column / row masks, causal, an all-masked
    row, backward, and the dropout output must drop exactly the elements the scalar kernel (f32 input) drops."""
    ops = _ops()
    from case_rg_amd import config
    O, R = 3, 50
    x = _rand(O, R, C, dt=torch.bfloat16, seed=1).requires_grad_()
    cv = torch.ones(O, C, dtype=torch.bool, device=DEV)
    cv[0, C // 2:] = False
    cv[2, 3] = False
    rv = torch.ones(O, R, dtype=torch.bool, device=DEV)
    rv[1, 40:] = False
    p = ops.masked_softmax(x, cv, rv, outer=O)
    xr = x.detach().float().requires_grad_()
    m = rv[:, :, None] & cv[:, None, :]
    pr = torch.softmax(xr.masked_fill(~m, float("-inf")), -1).masked_fill(~m, 0.0)
    _close(p, pr, 1e-2, "vec softmax p")
    assert (p[1, 40:] == 0).all(), "fully masked rows must be exactly zero"
    g = _rand(O, R, C, dt=torch.bfloat16, seed=2)
    p.backward(g)
    pr.backward(g.float())
    _close(x.grad, torch.nan_to_num(xr.grad), 2e-2, "vec softmax dx")
    if C >= R:  # causal: query r sees keys 0..r
        pc = ops.masked_softmax(x.detach(), None, None, outer=O, causal=True)
        tri = torch.ones(R, C, dtype=torch.bool, device=DEV).tril()
        prc = torch.softmax(x.detach().float().masked_fill(~tri, float("-inf")), -1)
        _close(pc, prc, 1e-2, "vec softmax causal")
    config.set_dropout(True)
    try:
        config.manual_seed(77)
        yv = ops.masked_softmax(x.detach(), cv, rv, outer=O, p_drop=0.3)
        config.manual_seed(77)
        ys = ops.masked_softmax(x.detach().float(), cv, rv, outer=O, p_drop=0.3)  # f32 input: scalar kernel, same counters
        live = pr.detach() > 1e-4
        assert ((yv == 0) == (ys == 0))[live].all(), "vector and scalar kernels must drop the same elements"
        kept = (yv != 0)[live].float().mean().item()
        assert 0.6 < kept < 0.8, kept
    finally:
        config.set_dropout(False)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_additive_scores(dt):
    ops = _ops()
    B, T, S, H = 2, 11, 150, 96
    wq = _rand(B, T, H, seed=1).requires_grad_()
    uh = _rand(B, S, H, dt=dt, seed=2).requires_grad_()
    v = _rand(H, seed=3).requires_grad_()
    s = ops.additive_scores(wq, uh, v)
    wr, ur, vr = [t.detach().float().requires_grad_() for t in (wq, uh, v)]
    sr = torch.tanh(wr[:, :, None, :] + ur[:, None, :, :]) @ vr
    tol = 1e-3 if dt == torch.float32 else 1e-2
    _close(s, sr, tol, "additive s")
    g = _rand(B, T, S, seed=4)
    s.backward(g)
    sr.backward(g)
    _close(wq.grad, wr.grad, tol, "additive dwq")
    _close(uh.grad, ur.grad, 3 * tol, "additive duh")
    _close(v.grad, vr.grad, tol, "additive dv")


def test_additive_scores_factored_exponential_regime():
    """The bf16 fast path of the additive attention factors 2^(w + u) = 2^w 2^u and clamps EACH prescaled argument at +-62 (csrc/attn_pointer.hip,
    ADVICE r5): inside |wq|, |uh| <= 21.4 the scores are tanh's to bf16 noise whatever the signs -- including arguments that nearly cancel --
    and beyond it the documented deviation appears: wq = 30, uh = -25 gives tanh(0) = 0 per feature instead of tanh(5) = 0.9999 (each factor
    saturates, their product is 2^0).  The reference's activations (LayerNorm outputs through a xavier Linear) stay within a few units, and the
    f32 parity path calls tanhf; this test pins where the boundary is, so that a change of the clamp shows up here."""
    ops = _ops()
    B, T, S, H = 1, 4, 64, 512
    v = torch.full((H,), 1.0 / H, device=DEV)
    for a, b, want in ((21.0, -20.0, math.tanh(1.0)), (-21.0, 20.5, math.tanh(-0.5)), (20.0, 21.0, 1.0), (-21.0, -21.0, -1.0), (3.0, -2.5, math.tanh(0.5))):
        wq = torch.full((B, T, H), a, device=DEV)
        uh = torch.full((B, S, H), b, device=DEV).to(torch.bfloat16)
        s = ops.additive_scores(wq, uh, v)
        ref = math.tanh(a + float(uh[0, 0, 0]))
        assert abs(ref - want) < 2e-2
        assert (s - ref).abs().max().item() <= 2e-3, (a, b, s.flatten()[0].item(), ref)
    wq = torch.full((B, T, H), 30.0, device=DEV)
    uh = torch.full((B, S, H), -25.0, device=DEV).to(torch.bfloat16)
    s = ops.additive_scores(wq, uh, v)
    assert (s.abs() <= 1e-3).all(), "outside the clamp the factored form saturates each factor: documented deviation (tanh(5) -> 0)"
    s32 = ops.additive_scores(wq, uh.float(), v)  # the f32 path is exact
    assert (s32 - math.tanh(5.0)).abs().max().item() <= 1e-5


def test_embed_pos_and_pointer_and_nll():
    ops = _ops()
    from oracle import sinusoid_table
    V, H, L = 50, 24, 9
    ids = torch.randint(0, V, (3, 2, L), generator=torch.Generator().manual_seed(1)).to(DEV)
    table = _rand(V, H, seed=2).requires_grad_()
    pe = sinusoid_table(20, H).to(DEV)
    y = ops.embed_pos(ids, table, pe, dtype=torch.float32)
    tr = table.detach().clone().requires_grad_()
    yr = F.embedding(ids, tr, padding_idx=0) * math.sqrt(H) + pe[:L]
    _close(y, yr, 1e-6, "embed y")
    g = _rand(3, 2, L, H, seed=3)
    y.backward(g)
    yr.backward(g)
    _close(table.grad, tr.grad, 1e-5, "embed dtable")
    # pointer scatter == dense one-hot bmm (common/Utils.py:344-355 + CaSE/Model.py:43)
    B, T, S = 2, 5, 17
    src = torch.randint(0, V, (B, S), generator=torch.Generator().manual_seed(4)).to(DEV)
    w = _rand(B, T, S, seed=5).abs().requires_grad_()
    d = ops.copy_scatter(src, w, V)
    wr = w.detach().clone().requires_grad_()
    dr = wr @ F.one_hot(src, V).float()
    _close(d, dr, 1e-5, "scatter dist")
    gd = _rand(B, T, V, seed=6)
    d.backward(gd)
    dr.backward(gd)
    _close(w.grad, wr.grad, 1e-6, "scatter dw")
    # nll with ignore_index 0
    dist = torch.softmax(_rand(B * T, V, seed=7), -1).requires_grad_()
    tgt = torch.randint(0, V, (B * T,), generator=torch.Generator().manual_seed(8)).to(DEV)
    tgt[0] = 0
    rows = ops.nll_rows(dist, tgt)
    dr2 = dist.detach().clone().requires_grad_()
    ref = F.nll_loss((dr2 + 1e-8).log(), tgt, ignore_index=0, reduction="none")
    _close(rows, ref, 1e-5, "nll rows")
    rows.sum().backward()
    ref.sum().backward()
    _close(dist.grad, dr2.grad, 1e-5, "nll ddist")
    # argmax ties -> lowest index
    x = torch.tensor([[0.1, 0.7, 0.7, 0.05], [0.3, 0.3, 0.2, 0.3]], device=DEV)
    idx, val = ops.row_argmax(x)
    assert idx.tolist() == [1, 0] and val.tolist() == pytest.approx([0.7, 0.3])


def test_dropout_mask_is_regenerated_in_backward():
    from case_rg_amd import config, ops
    config.set_dropout(True)
    config.manual_seed(7)
    try:
        x = torch.ones(1000, 64, device=DEV, requires_grad=True)
        y = ops.dropout(x, 0.1, training=True)
        keep = (y != 0)
        assert 0.85 < keep.float().mean().item() < 0.95
        _close(y[keep], torch.full_like(y[keep], 1 / 0.9), 1e-6, "dropout scale")
        y.sum().backward()
        assert torch.equal(x.grad != 0, keep), "backward must regenerate the same mask"
        # GEMM-epilogue dropout: same statistics, backward consistent with forward mask
        w = torch.eye(64, device=DEV).requires_grad_()
        x2 = torch.ones(512, 64, device=DEV, requires_grad=True)
        z = ops.linear(x2, w, None, p_drop=0.25)
        kz = z != 0
        assert 0.70 < kz.float().mean().item() < 0.80
        z.sum().backward()
        assert torch.equal(x2.grad != 0, kz)
    finally:
        config.set_dropout(False)


def test_gemm_rejects_bad_arguments():
    from case_rg_amd import ops
    a = torch.zeros(4, 4, device=DEV)
    with pytest.raises(RuntimeError, match="case_gemm"):
        ops.gemm(a, a, a, 0, 4, 4, 4, 4, 4)
    with pytest.raises(RuntimeError, match="GPU only"):
        ops.linear(torch.zeros(2, 2), torch.zeros(2, 2))


@pytest.mark.parametrize("N,h,Lq,Lk,d,causal", [(2, 8, 384, 384, 64, False), (2, 8, 384, 384, 320, False), (3, 8, 40, 40, 64, True),
                                                 (2, 8, 40, 520, 64, False), (2, 2, 200, 333, 320, False),
                                                 (2, 8, 512, 512, 96, False), (2, 4, 512, 512, 480, False), (2, 3, 70, 70, 96, True),
                                                 (1, 2, 100, 45, 480, False), (1, 8, 40, 4200, 96, False), (2, 2, 33, 2500, 320, False),
                                                 # the reference's default width 256 (CaSE/Run.py:72-78): head_dim 32 in the H-wide stacks, 160 in the 5H blocks;
                                                 # passages of 100, queries of 60, answers of 40 (causal), the 1000-token memory
                                                 (3, 8, 100, 100, 32, False), (2, 8, 40, 40, 32, True), (2, 8, 40, 1000, 32, False), (2, 8, 60, 60, 32, False),
                                                 (3, 8, 100, 100, 160, False), (2, 8, 60, 60, 160, False), (1, 8, 40, 2100, 32, False), (2, 4, 130, 70, 160, False)])
def test_fused_attention_matches_unfused_and_reference(N, h, Lq, Lk, d, causal):
    """bf16 fused kernel (no score tensor) vs the f32 reference and vs the unfused GEMM+softmax path, incl. identical
    dropout masks (same counter RNG / element index)."""
    from case_rg_amd import _abi, config
    ops = _ops()
    assert _abi.lib.case_attention_supported(d)
    ops.ATTENTION_MODE = "fused"  # also head sizes whose backward is not fused: fused forward + recompute backward under test
    try:
        _fused_attention_case(ops, _abi, config, N, h, Lq, Lk, d, causal)
    finally:
        ops.ATTENTION_MODE = "auto"


def _fused_attention_case(ops, _abi, config, N, h, Lq, Lk, d, causal):
    E, dt = h * d, torch.bfloat16
    valid = torch.ones(N, Lk, dtype=torch.bool, device=DEV)
    valid[N - 1, Lk // 2 + 3:] = False
    self_attn = Lq == Lk
    if self_attn:
        src = _rand(N, Lq, 3 * E, dt=dt, seed=1, scale=0.7).requires_grad_()
        args = (src, src, src, 0, E, 2 * E)
        q, k, v = src.detach().float().split(E, dim=-1)
    else:
        qs = _rand(N, Lq, E, dt=dt, seed=1, scale=0.7).requires_grad_()
        kv = _rand(N, Lk, 2 * E, dt=dt, seed=2, scale=0.7).requires_grad_()
        args = (qs, kv, kv, 0, 0, E)
        q, (k, v) = qs.detach().float(), kv.detach().float().split(E, dim=-1)
    o = ops.attention(*args, h, d, key_valid=valid, causal=causal)
    qh, kh, vh = [t.reshape(N, -1, h, d).transpose(1, 2) for t in (q, k, v)]
    s = (qh @ kh.transpose(-1, -2)) / math.sqrt(d)
    s = s.masked_fill(~valid[:, None, None, :], float("-inf"))
    if causal:
        s = s + torch.triu(torch.full((Lq, Lk), float("-inf"), device=DEV), 1)
    ref = (torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(N, Lq, E)
    _close(o, ref, 2e-2, "fused attn o")
    g = _rand(N, Lq, E, dt=dt, seed=3)
    o.backward(g)
    # gradients against the f32 reference of the same (bf16-rounded) inputs
    leaves = [t for t in (args[0], args[1]) if t.requires_grad]
    leaves = [leaves[0]] if self_attn else leaves
    refs = [t.detach().float().requires_grad_() for t in leaves]
    if self_attn:
        rq, rk, rv = refs[0].split(E, dim=-1)
    else:
        rq, (rk, rv) = refs[0], refs[1].split(E, dim=-1)
    rqh, rkh, rvh = [t.reshape(N, -1, h, d).transpose(1, 2) for t in (rq, rk, rv)]
    rs = (rqh @ rkh.transpose(-1, -2)) / math.sqrt(d)
    rs = rs.masked_fill(~valid[:, None, None, :], float("-inf"))
    if causal:
        rs = rs + torch.triu(torch.full((Lq, Lk), float("-inf"), device=DEV), 1)
    (torch.softmax(rs, -1) @ rvh).transpose(1, 2).reshape(N, Lq, E).backward(g.float())
    for a_, r_ in zip(leaves, refs):
        _close(a_.grad, r_.grad, 4e-2, "fused attn grad")
    # dropout: fused forward must equal the unfused path bit-for-bit in its mask (compare through the outputs)
    config.set_dropout(True)
    try:
        config.manual_seed(11)
        o_f = ops.attention(*[a.detach() if torch.is_tensor(a) else a for a in args], h, d, key_valid=valid, causal=causal, p_drop=0.1)
        config.manual_seed(11)
        saved = _abi.lib.case_attention_supported
        try:
            _abi.lib.case_attention_supported = lambda _d: 0
            o_u = ops.attention(*[a.detach() if torch.is_tensor(a) else a for a in args], h, d, key_valid=valid, causal=causal, p_drop=0.1)
        finally:
            _abi.lib.case_attention_supported = saved
        _close(o_f, o_u, 2e-2, "fused vs unfused with dropout")
        # and the fused backward (where built) regenerates that same mask: gradients agree with the unfused path
        if _abi.lib.case_attention_bwd_supported(d):
            grads = []
            for fused_on in (True, False):
                config.manual_seed(11)
                ins = [a.detach().clone().requires_grad_() if torch.is_tensor(a) else a for a in args]
                if self_attn:
                    ins[1] = ins[2] = ins[0]
                else:
                    ins[2] = ins[1]
                saved = _abi.lib.case_attention_supported
                try:
                    if not fused_on:
                        _abi.lib.case_attention_supported = lambda _d: 0
                    ops.attention(*ins, h, d, key_valid=valid, causal=causal, p_drop=0.1).backward(g)
                finally:
                    _abi.lib.case_attention_supported = saved
                grads.append([t.grad for t in (ins[0], ins[1]) if t.grad is not None])
            for gf, gu in zip(*grads):
                _close(gf, gu, 4e-2, "fused vs unfused gradients with dropout")
    finally:
        config.set_dropout(False)


@pytest.mark.parametrize("N,ragged", [(32, False), (4, True)])
def test_decoder_cross_attention_at_the_benchmark_memory_length(N, ragged):
    """The decoder's cross-attention as bench.py times it (cfg 2: T = 40 queries x 8 heads of 64 over the 10 x 384 = 3840-token memory,
    common/TransformerDecoder.py:81-82, CaSE/Model.py:57-58): ops._kv_splits sends its forward to case_attention_fwd_splitkv and its
    backward to the merged dQ | dK / dV launch.  Values AND gradients against an f32 reference on the same bf16 inputs, full batch and
    a ragged one (keys masked per passage as padded passages are, one item WITHOUT any valid key: exact zeros there)."""
    from case_rg_amd import _abi
    ops = _ops()
    h, Lq, Lk, d = 8, 40, 3840, 64
    E, dt = h * d, torch.bfloat16
    valid = torch.ones(N, Lk, dtype=torch.bool, device=DEV)
    if ragged:
        lens = torch.tensor([[384, 200, 2, 377, 192, 384, 300, 250, 2, 311]], device=DEV).expand(N, 10).clone()
        lens[1] = torch.tensor([2, 2, 2, 384, 2, 2, 2, 2, 2, 193], device=DEV)
        valid = (torch.arange(384, device=DEV)[None, None, :] < lens[:, :, None]).reshape(N, Lk)
        valid[2] = False  # no key at all
    qs = _rand(N, Lq, E, dt=dt, seed=1, scale=0.7).requires_grad_()
    kv = _rand(N, Lk, 2 * E, dt=dt, seed=2, scale=0.7).requires_grad_()
    seen, raw = [], _abi.call

    def spy(name, *a):
        if name.startswith("case_attention"):
            seen.append((name, int(a[0].head_dim), int(a[0].Lk)))
        return raw(name, *a)

    _abi.call = spy
    try:
        assert ops._kv_splits(N, h, Lq, Lk, False) > 1
        o = ops.attention(qs, kv, kv, 0, 0, E, h, d, key_valid=valid, causal=False)
        g = _rand(N, Lq, E, dt=dt, seed=3)
        o.backward(g)
        torch.cuda.synchronize()
    finally:
        _abi.call = raw
    assert ("case_attention_fwd_splitkv", 64, Lk) in seen, seen
    assert ("case_attention_bwd", 64, Lk) in seen, seen
    rq, rkv = qs.detach().float().requires_grad_(), kv.detach().float().requires_grad_()
    rk, rv = rkv.split(E, dim=-1)
    rqh, rkh, rvh = [t.reshape(N, -1, h, d).transpose(1, 2) for t in (rq, rk, rv)]
    sc = (rqh @ rkh.transpose(-1, -2)) / math.sqrt(d)
    sc = sc.masked_fill(~valid[:, None, None, :], float("-inf"))
    p = torch.nan_to_num(torch.softmax(sc, -1), nan=0.0)  # a sequence without a valid key: zeros, as the reference's callers mask it
    ref = (p @ rvh).transpose(1, 2).reshape(N, Lq, E)
    ref.backward(g.float())
    _close(o, ref, 2e-2, "cross-attention o")
    _close(qs.grad, rq.grad, 4e-2, "cross-attention dq")
    _close(kv.grad, rkv.grad, 4e-2, "cross-attention dkv")
    if ragged:
        assert float(o[2].float().abs().max()) == 0.0 and float(qs.grad[2].float().abs().max()) == 0.0
        assert float(kv.grad[2].float().abs().max()) == 0.0
        assert float(kv.grad.float().masked_select(~valid[:, :, None]).abs().max()) == 0.0, "gradient on a masked key"


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_gemm_k_tile_counts(dt):
    """Every K-loop shape of the software pipeline: 1, 2, 3 (tail paths), 4, 5, 7 and 16 K tiles, full and edge tiles."""
    ops = _ops()
    per_tile = 32 if dt == torch.float32 else 64
    for tiles in (1, 2, 3, 4, 5, 7, 16):
        for (M, N) in ((128, 128), (16, 192), (200, 130)):
            K = tiles * per_tile
            x = _rand(M, K, dt=dt, seed=tiles)
            w = _rand(N, K, seed=100 + tiles, scale=K ** -0.5)
            y = ops.linear(x, w, None)
            ref = F.linear(x.float(), w.to(dt).float())
            _close(y, ref, _tol(dt), "gemm K=%d M=%d N=%d" % (K, M, N))


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_linear_carry_merges_both_gradients_of_x(dt):
    """x + f(linear(x)): with linear_carry the residual's gradient is added in the dX GEMM epilogue."""
    ops = _ops()
    M, K, N = 130, 64, 192
    x = _rand(M, K, dt=dt, seed=1).requires_grad_()
    w, b = _rand(N, K, seed=2, scale=K ** -0.5).requires_grad_(), _rand(N, seed=3).requires_grad_()
    y, xc = ops.linear_carry(x, w, b)
    out = y[:, :K] * 0.5 + xc
    g = _rand(M, K, dt=dt, seed=4)
    out.backward(g)
    xr = x.detach().float().requires_grad_()
    wr, br = w.detach().to(dt).float().requires_grad_(), b.detach().float().requires_grad_()
    outr = F.linear(xr, wr, br)[:, :K] * 0.5 + xr
    outr.backward(g.float())
    _close(out, outr, _tol(dt), "carry out")
    _close(x.grad, xr.grad, 2 * _tol(dt), "carry dx")
    _close(w.grad, wr.grad, 2 * _tol(dt), "carry dw")


def test_linear_wide_output_backward_splits_reduction():
    """Input gradient of a vocabulary-sized projection (N >= 4096, few output tiles) takes the split-K f32 path."""
    ops = _ops()
    M, K, N = 200, 128, 4224
    x = _rand(M, K, dt=torch.bfloat16, seed=1).requires_grad_(True)
    w = torch.nn.Parameter(_rand(N, K, seed=2, scale=K ** -0.5))
    y = ops.linear(x, w, None)
    g = _rand(M, N, dt=torch.bfloat16, seed=3, scale=0.1)
    y.backward(g)
    xr = x.detach().float().requires_grad_(True)
    wr = w.detach().to(torch.bfloat16).float().requires_grad_(True)
    F.linear(xr, wr).backward(g.float())
    _close(x.grad, xr.grad, 2e-2, "split-K dX")
    _close(w.grad, wr.grad, 2e-2, "dW next to split-K dX")


class _Tile:
    """CaseGemmDesc.tile for the duration of a block (128 = 128x128 only, 256 = 256x256 whenever eligible)."""

    def __init__(self, tile):
        self.tile = tile

    def __enter__(self):
        from case_rg_amd import ops
        ops.GEMM_TILE = self.tile

    def __exit__(self, *exc):
        from case_rg_amd import ops
        ops.GEMM_TILE = 0


def _gemm_both_tilings(fn):
    with _Tile(128):
        small = fn()
    with _Tile(256):
        large = fn()
    torch.cuda.synchronize()
    return small, large


@pytest.mark.parametrize("layout", ["nt", "nn", "tn"])
def test_gemm_large_tile_layouts(layout):
    """256x256 tiling against the 128x128 tiling and torch: 1..5 K tiles (pipeline prologue / tail), several output tiles."""
    ops = _ops()
    dt = torch.bfloat16
    for (M, N, K) in ((256, 256, 64), (512, 768, 128), (256, 512, 192), (768, 256, 320)):
        if layout == "nt":      # C = A[M,K] . B[N,K]^T
            a, b = _rand(M, K, dt=dt, seed=1), _rand(N, K, dt=dt, seed=2, scale=K ** -0.5)
            ref = a.float() @ b.float().t()
            fn = lambda: ops.gemm(a, b, torch.empty(M, N, device="cuda", dtype=dt), M, N, K, K, K, N)
        elif layout == "nn":    # C = A[M,K] . B[K,N]
            a, b = _rand(M, K, dt=dt, seed=1), _rand(K, N, dt=dt, seed=2, scale=K ** -0.5)
            ref = a.float() @ b.float()
            fn = lambda: ops.gemm(a, b, torch.empty(M, N, device="cuda", dtype=dt), M, N, K, K, N, N, b_kmajor=True)
        else:                   # C = A[K,M]^T . B[K,N]
            a, b = _rand(K, M, dt=dt, seed=1), _rand(K, N, dt=dt, seed=2, scale=K ** -0.5)
            ref = a.float().t() @ b.float()
            fn = lambda: ops.gemm(a, b, torch.empty(M, N, device="cuda", dtype=dt), M, N, K, M, N, N, a_kmajor=True,
                                  b_kmajor=True)
        small, large = _gemm_both_tilings(fn)
        _close(large, ref, 2e-2, "large tile %s M=%d N=%d K=%d" % (layout, M, N, K))
        _close(large, small.float(), 1e-2, "tilings agree %s M=%d N=%d K=%d" % (layout, M, N, K))


def test_gemm_large_tile_epilogues_and_persistence():
    """Fused epilogues of the 256x256 tiling (bias + GELU with the saved pre-activation, dropout + residual with the same
    keep mask as the 128x128 tiling, f32 output) and more tiles than CUs (workgroups walk several tiles)."""
    ops = _ops()
    from case_rg_amd import _abi as A
    dt = torch.bfloat16
    M, N, K = 512, 512, 128
    x, w = _rand(M, K, dt=dt, seed=1), _rand(N, K, dt=dt, seed=2, scale=K ** -0.5)
    bias, res = _rand(N, seed=3), _rand(M, N, dt=dt, seed=4)

    def gelu():
        y, pre = torch.empty(M, N, device="cuda", dtype=dt), torch.empty(M, N, device="cuda", dtype=dt)
        ops.gemm(x, w, y, M, N, K, K, K, N, epilogue=A.EPI_BIAS_COL | A.EPI_GELU, bias_col=bias, aux_out=pre, ld_aux=N)
        return torch.stack([y.float(), pre.float()])
    small, large = _gemm_both_tilings(gelu)
    pre_ref = x.float() @ w.float().t() + bias
    _close(large[1], pre_ref, 2e-2, "large tile pre-activation")
    _close(large[0], F.gelu(pre_ref), 2e-2, "large tile gelu")
    _close(large, small, 1e-2, "gelu epilogue: tilings agree")

    def drop_res():
        y = torch.empty(M, N, device="cuda", dtype=dt)
        ops.gemm(x, w, y, M, N, K, K, K, N, epilogue=A.EPI_BIAS_COL | A.EPI_RESIDUAL, bias_col=bias, aux=res, ld_aux=N,
                 drop=(0.25, 1234, 77))
        return y.float()
    small, large = _gemm_both_tilings(drop_res)
    _close(large, small, 1e-2, "dropout + residual: same keep mask in both tilings")
    kept = ((large - res.float()).abs() > 1e-6).float().mean().item()
    assert 0.70 < kept < 0.80, kept

    def f32_out():
        y = torch.empty(M, N, device="cuda", dtype=torch.float32)
        return ops.gemm(x, w, y, M, N, K, K, K, N, alpha=0.5)
    small, large = _gemm_both_tilings(f32_out)
    _close(large, 0.5 * (x.float() @ w.float().t()), 1e-3, "large tile f32 out")

    # split-K weight gradient (TN, f32 atomics) with uneven splits
    Kl = 64 * 11
    g, xx = _rand(Kl, M, dt=dt, seed=5), _rand(Kl, N, dt=dt, seed=6, scale=Kl ** -0.5)

    def dw():
        out = torch.zeros(M, N, device="cuda", dtype=torch.float32)
        return ops.gemm(g, xx, out, M, N, Kl, M, N, N, a_kmajor=True, b_kmajor=True, split_k=4, epilogue=A.EPI_ATOMIC)
    small, large = _gemm_both_tilings(dw)
    _close(large, g.float().t() @ xx.float(), 2e-3, "large tile split-K")

    # 20 x 16 = 320 output tiles on 256 CUs: the persistent loop and the cross-tile prefetch
    M2, N2, K2 = 5120, 4096, 192
    a, b = _rand(M2, K2, dt=dt, seed=7), _rand(N2, K2, dt=dt, seed=8, scale=K2 ** -0.5)
    small, large = _gemm_both_tilings(lambda: ops.gemm(a, b, torch.empty(M2, N2, device="cuda", dtype=dt), M2, N2, K2, K2, K2, N2))
    _close(large, a.float() @ b.float().t(), 2e-2, "persistent large tile")
    _close(large, small.float(), 1e-2, "persistent: tilings agree")


@pytest.mark.parametrize("layout", ["nt", "nn", "tn"])
def test_gemm_small_tile_layouts_and_epilogues(layout):
    """64x64 tiling (whole K panel in LDS) against the 128x128 tiling and torch: 1..10 K tiles, every operand layout, the
    epilogue words of the decoder-side projections, split-K atomics."""
    ops = _ops()
    from case_rg_amd import _abi as A
    dt = torch.bfloat16

    def both(fn):
        with _Tile(128):
            small = fn()
        with _Tile(64):
            tiny = fn()
        torch.cuda.synchronize()
        return small, tiny

    for (M, N, K) in ((64, 64, 64), (256, 512, 512), (1280, 512, 512), (128, 192, 640), (320, 64, 192)):
        if layout == "nt":
            a, b = _rand(M, K, dt=dt, seed=1), _rand(N, K, dt=dt, seed=2, scale=K ** -0.5)
            ref = a.float() @ b.float().t()
            fn = lambda: ops.gemm(a, b, torch.full((M, N), float("nan"), device="cuda", dtype=dt), M, N, K, K, K, N)
        elif layout == "nn":
            a, b = _rand(M, K, dt=dt, seed=1), _rand(K, N, dt=dt, seed=2, scale=K ** -0.5)
            ref = a.float() @ b.float()
            fn = lambda: ops.gemm(a, b, torch.full((M, N), float("nan"), device="cuda", dtype=dt), M, N, K, K, N, N, b_kmajor=True)
        else:
            a, b = _rand(K, M, dt=dt, seed=1), _rand(K, N, dt=dt, seed=2, scale=K ** -0.5)
            ref = a.float().t() @ b.float()
            fn = lambda: ops.gemm(a, b, torch.full((M, N), float("nan"), device="cuda", dtype=dt), M, N, K, M, N, N, a_kmajor=True,
                                  b_kmajor=True)
        small, tiny = both(fn)
        _close(tiny, ref, 2e-2, "small tile %s M=%d N=%d K=%d" % (layout, M, N, K))
        _close(tiny, small.float(), 1e-2, "tilings agree %s M=%d N=%d K=%d" % (layout, M, N, K))
    if layout != "nt":
        return
    M, N, K = 256, 512, 512
    x, w = _rand(M, K, dt=dt, seed=1), _rand(N, K, dt=dt, seed=2, scale=K ** -0.5)
    bias, res = _rand(N, seed=3), _rand(M, N, dt=dt, seed=4)

    def gelu():
        y, pre = torch.empty(M, N, device="cuda", dtype=dt), torch.empty(M, N, device="cuda", dtype=dt)
        ops.gemm(x, w, y, M, N, K, K, K, N, epilogue=A.EPI_BIAS_COL | A.EPI_GELU, bias_col=bias, aux_out=pre, ld_aux=N)
        return torch.stack([y.float(), pre.float()])
    small, tiny = both(gelu)
    _close(tiny[0], F.gelu(x.float() @ w.float().t() + bias), 2e-2, "small tile gelu")
    _close(tiny, small, 1e-2, "gelu epilogue: tilings agree")

    def drop_res():
        y = torch.empty(M, N, device="cuda", dtype=dt)
        ops.gemm(x, w, y, M, N, K, K, K, N, epilogue=A.EPI_BIAS_COL | A.EPI_RESIDUAL, bias_col=bias, aux=res, ld_aux=N,
                 drop=(0.25, 1234, 77))
        return y.float()
    small, tiny = both(drop_res)
    _close(tiny, small, 1e-2, "dropout + residual: same keep mask in both tilings")

    def f32_out():
        return ops.gemm(x, w, torch.empty(M, N, device="cuda", dtype=torch.float32), M, N, K, K, K, N, alpha=0.5)
    small, tiny = both(f32_out)
    _close(tiny, 0.5 * (x.float() @ w.float().t()), 1e-3, "small tile f32 out")
    Kl = 64 * 11  # split-K weight gradient with uneven splits (4, 4, 3 K tiles)
    g, xx = _rand(Kl, M, dt=dt, seed=5), _rand(Kl, N, dt=dt, seed=6, scale=Kl ** -0.5)

    def dw():
        out = torch.zeros(M, N, device="cuda", dtype=torch.float32)
        return ops.gemm(g, xx, out, M, N, Kl, M, N, N, a_kmajor=True, b_kmajor=True, split_k=3, epilogue=A.EPI_ATOMIC)
    small, tiny = both(dw)
    _close(tiny, g.float().t() @ xx.float(), 2e-3, "small tile split-K")
    with _Tile(64):  # more than 10 K tiles per split: not eligible, the call falls back to 128x128 (and stays correct)
        big = ops.gemm(_rand(64, 1024, dt=dt, seed=7), _rand(64, 1024, dt=dt, seed=8), torch.empty(64, 64, device="cuda", dtype=dt),
                       64, 64, 1024, 1024, 1024, 64)
    _close(big, _rand(64, 1024, dt=dt, seed=7).float() @ _rand(64, 1024, dt=dt, seed=8).float().t(), 2e-2, "fallback")


@pytest.mark.parametrize("tile", [64, 256])
def test_gemm_dma_tilings_with_padded_leading_dimensions(tile):
    """The LDS-DMA tilings address operands through (row * ld) byte offsets: operands and C that are column blocks of wider
    buffers (ld > extent), NT and the k-major layouts, aux with its own ld."""
    ops = _ops()
    from case_rg_amd import _abi as A
    dt = torch.bfloat16
    M, N, K = (512, 256, 320) if tile == 256 else (128, 192, 320)
    big_a, big_b = _rand(M, K + 64, dt=dt, seed=1), _rand(N, K + 128, dt=dt, seed=2, scale=K ** -0.5)
    a, b = big_a[:, 64:], big_b[:, 128:]                      # row-major views: lda = K + 64, ldb = K + 128
    big_c = torch.full((M, N + 64), float("nan"), device="cuda", dtype=dt)
    big_r = _rand(M, N + 32, dt=dt, seed=3)
    bias = _rand(N, seed=4)
    with _Tile(tile):
        ops.gemm(big_a, big_b, big_c, M, N, K, K + 64, K + 128, N + 64, a_off=64, b_off=128, c_off=64, epilogue=A.EPI_BIAS_COL | A.EPI_RESIDUAL,
                 bias_col=bias, aux=big_r, ld_aux=N + 32)
    torch.cuda.synchronize()
    ref = a.float() @ b.float().t() + bias + big_r[:, :N].float()
    _close(big_c[:, 64:], ref, 2e-2, "padded NT tile %d" % tile)
    assert torch.isnan(big_c[:, :64].float()).all(), "columns outside the C block untouched"
    # k-major operands inside wider buffers (the weight-gradient layout)
    ga, gb = _rand(K, M + 64, dt=dt, seed=5), _rand(K, N + 64, dt=dt, seed=6, scale=K ** -0.5)
    out = torch.zeros(M, N, device="cuda", dtype=torch.float32)
    with _Tile(tile):
        ops.gemm(ga, gb, out, M, N, K, M + 64, N + 64, N, a_off=64, b_off=0, a_kmajor=True, b_kmajor=True, split_k=1, epilogue=A.EPI_ATOMIC)
    torch.cuda.synchronize()
    _close(out, ga[:, 64:].float().t() @ gb[:, :N].float(), 2e-3, "padded TN tile %d" % tile)


@pytest.mark.parametrize("layout", ["nt", "nn"])
def test_gemm_large_tile_is_bit_reproducible(layout):
    """Race screen of the LDS-DMA pipeline (counted waits, barriers, stage re-use, strips under the next tile's prologue): the
    non-atomic kernels have a fixed summation order, so repeated launches must agree bit for bit -- on an output prefilled with
    NaN, with more tiles than CUs (persistent loop) and an odd number of K tiles."""
    ops = _ops()
    from case_rg_amd import _abi as A
    dt = torch.bfloat16
    M, N, K = 5120, 4352, 448  # 20 x 17 = 340 tiles on 256 CUs, 7 K tiles
    a = _rand(M, K, dt=dt, seed=1)
    b = _rand(N, K, dt=dt, seed=2, scale=K ** -0.5) if layout == "nt" else _rand(K, N, dt=dt, seed=2, scale=K ** -0.5)
    bias, res = _rand(N, seed=3), _rand(M, N, dt=dt, seed=4)
    first = None
    with _Tile(256):
        for rep in range(12):
            c = torch.full((M, N), float("nan"), device="cuda", dtype=dt)
            if layout == "nt":
                ops.gemm(a, b, c, M, N, K, K, K, N, epilogue=A.EPI_BIAS_COL | A.EPI_RESIDUAL, bias_col=bias, aux=res, ld_aux=N)
            else:
                ops.gemm(a, b, c, M, N, K, K, N, N, b_kmajor=True, epilogue=A.EPI_BIAS_COL | A.EPI_RESIDUAL, bias_col=bias, aux=res, ld_aux=N)
            if first is None:
                first = c
                ref = (a.float() @ (b.float().t() if layout == "nt" else b.float())) + bias + res.float()
                _close(c, ref, 2e-2, "reference %s" % layout)
            else:
                assert torch.equal(c, first), "launch %d differs from launch 0 (%s)" % (rep, layout)


def test_weight_gradient_gemm_also_sums_the_bias_gradient():
    """case_gemm_dw_bias: dW = dY^T X with the bias gradient (column sums of dY) taken from the k-major A fragments of the same
    launch (256x256 tiling), and the fallback (separate column-sum pass) for calls the large tiling does not take."""
    ops = _ops()
    from case_rg_amd import _abi as A
    dt = torch.bfloat16
    for (Mtok, N, K, split) in ((64 * 24, 512, 256, 3), (64 * 40, 768, 512, 5), (64 * 7, 256, 256, 1), (64 * 9, 320, 256, 2)):
        g, x = _rand(Mtok, N, dt=dt, seed=1), _rand(Mtok, K, dt=dt, seed=2, scale=Mtok ** -0.5)
        dw = torch.zeros(N, K, device="cuda", dtype=torch.float32)
        db = torch.zeros(N, device="cuda", dtype=torch.float32)
        ops.gemm(g, x, dw, N, K, Mtok, N, K, K, a_kmajor=True, b_kmajor=True, split_k=split, epilogue=A.EPI_ATOMIC, rowsum_out=db)
        torch.cuda.synchronize()
        _close(dw, g.float().t() @ x.float(), 2e-3, "dW with fused bias gradient N=%d" % N)
        _close(db, g.float().sum(0), 2e-3, "bias gradient N=%d (fused on the 256 tiling: %s)" % (N, N % 256 == 0))
    with pytest.raises(ValueError):
        ops.gemm(g, x, dw, 320, 256, 576, 320, 256, 256, b_kmajor=True, epilogue=A.EPI_ATOMIC, rowsum_out=db)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cols_k", [1, 2, 5, 0.5, 1.5, 2.5, 7.5])
def test_layernorm_backward_with_the_dropout_masked_copy_matches_two_passes_bit_for_bit(dt, cols_k):
    """case_layernorm_bwd_dropout: dx and mask * dx / (1 - p) from one kernel == case_layernorm_bwd followed by case_dropout (same bits:
    the mask is applied to the rounded dx), for the one-wave-per-row kernels (k = 1, 2) and the row-split kernel (k = 5: the 5H rows)."""
    ops = _ops()
    from case_rg_amd import _abi as A
    if dt == torch.float32 and cols_k != int(cols_k):
        pytest.skip("the 8-byte-vector rows are a bf16 form")
    rows, cols = 77, int(cols_k * (256 if dt == torch.float32 else 512))  # bf16 x 0.5 / 1.5 / 2.5 / 7.5 = 256 / 768 / 1280 / 3840: the 8-byte-vector kernels
    x, dy = _rand(rows, cols, dt=dt, seed=1), _rand(rows, cols, dt=dt, seed=2)
    gamma = _rand(cols, seed=3) + 1.0
    mean, rstd = x.float().mean(1).contiguous(), (x.float().var(1, unbiased=False) + 1e-5).rsqrt().contiguous()
    code = A.F32 if dt == torch.float32 else A.BF16
    p, seed, off = 0.1, 1234567, 4242
    dx0, dg0, db0 = torch.empty_like(x), torch.zeros(cols, device="cuda"), torch.zeros(cols, device="cuda")
    A.call("case_layernorm_bwd", dy.data_ptr(), x.data_ptr(), None, gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dx0.data_ptr(), None,
           dg0.data_ptr(), db0.data_ptr(), rows, cols, code, 0)
    g0 = torch.empty_like(x)
    A.call("case_dropout", dx0.data_ptr(), g0.data_ptr(), x.numel(), p, seed, off, None, code, 0)
    dx1, g1, dg1, db1 = torch.empty_like(x), torch.empty_like(x), torch.zeros(cols, device="cuda"), torch.zeros(cols, device="cuda")
    A.call("case_layernorm_bwd_dropout", dy.data_ptr(), x.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dx1.data_ptr(),
           g1.data_ptr(), dg1.data_ptr(), db1.data_ptr(), rows, cols, p, seed, off, None, code, 0)
    torch.cuda.synchronize()
    if cols_k == 5 and dt == torch.bfloat16:
        # the 5H rows take the one-wave-per-row kernel: its row sums are formed in another order than the row-split kernel's, so dx may
        # differ in the last bf16 bit; the masked copy must still be case_dropout of ITS dx, bit for bit
        _close(dx1, dx0, 8e-3, "dx (one wave per row)")
        A.call("case_dropout", dx1.data_ptr(), g0.data_ptr(), x.numel(), p, seed, off, None, code, 0)
        torch.cuda.synchronize()
    else:
        assert torch.equal(dx0, dx1), "dx differs"
    assert torch.equal(g0, g1), "masked copy differs"
    assert 0.05 < (g1 == 0).float().mean().item() < 0.16
    _close(dg1, dg0, 1e-5, "d_gamma")
    _close(db1, db0, 1e-5, "d_beta")
    with pytest.raises(RuntimeError, match="64-lane"):
        A.call("case_layernorm_bwd_dropout", dy.data_ptr(), x.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dx1.data_ptr(),
               g1.data_ptr(), dg1.data_ptr(), db1.data_ptr(), rows, cols - 8, p, seed, off, None, code, 0)


@pytest.mark.parametrize("form", ["self", "cross", "key_is_not_value"])
def test_multihead_attention_general_call_forms_match_torch(form):
    """The nn.MultiheadAttention-compatible ``forward`` (sequence-first): output AND head-averaged weights (``need_weights``) against
    torch.nn.MultiheadAttention itself in eval mode, for self-attention with the causal mask, cross-attention with key padding, and the
    general form with different key and value tensors (reference call sites: common/TransformerDecoder.py:77-82)."""
    from case_rg_amd.common.attention import MultiheadAttention
    _ops()
    E, h, N, Lq, Lk = 128, 4, 3, 10, 14
    mha = MultiheadAttention(E, h, dropout=0.1).to(DEV).eval()
    ref = torch.nn.MultiheadAttention(E, h, dropout=0.1).eval()
    with torch.no_grad():
        mha.in_proj_bias.copy_(_rand(3 * E, seed=4) * 0.1)
        mha.out_proj.bias.copy_(_rand(E, seed=5) * 0.1)
        ref.in_proj_weight.copy_(mha.in_proj_weight.cpu())
        ref.in_proj_bias.copy_(mha.in_proj_bias.cpu())
        ref.out_proj.weight.copy_(mha.out_proj.weight.cpu())
        ref.out_proj.bias.copy_(mha.out_proj.bias.cpu())
    q = _rand(Lq, N, E, seed=1)
    if form == "self":
        k = v = q
        pad = torch.zeros(N, Lq, dtype=torch.bool)
        mask = torch.triu(torch.full((Lq, Lq), float("-inf")), 1)
    else:
        k = _rand(Lk, N, E, seed=2)
        v = k if form == "cross" else _rand(Lk, N, E, seed=3)
        pad = torch.zeros(N, Lk, dtype=torch.bool)
        pad[1, 9:] = True
        mask = None
    with torch.no_grad():
        out, w = mha(q, k, v, attn_mask=None if mask is None else mask.to(DEV), key_padding_mask=pad.to(DEV), need_weights=True)
        out2, w2 = mha(q, k, v, attn_mask=None if mask is None else mask.to(DEV), key_padding_mask=pad.to(DEV))
        want, ww = ref(q.cpu(), k.cpu(), v.cpu(), attn_mask=mask, key_padding_mask=pad, need_weights=True)
    assert w2 is None and torch.equal(out, out2)
    _close(out.cpu(), want, 1e-4, "attention output (%s)" % form)
    _close(w.cpu(), ww, 1e-4, "head-averaged weights (%s)" % form)


def test_decoder_stack_returns_the_last_layers_attention_weights_on_request():
    """``return_attention = True`` on a TransformerDecoder: (output, self-attention weights, memory-attention weights) of the LAST layer,
    head-averaged, as the reference returns them (common/TransformerDecoder.py:77-90, :208-218) -- against the f32 oracle; off by
    default (None, None)."""
    import case_rg_amd
    import oracle
    from case_rg_amd.utils import fill_params
    _ops()
    ns = case_rg_amd.namespace()
    E, h, N, T, S, layers = 128, 4, 3, 9, 21, 2
    dec = fill_params(ns.TransformerDecoder(ns.TransformerDecoderLayer(E, h, dim_feedforward=E, dropout=0.1, activation="gelu"), layers), 5).to(DEV).eval()
    ref = fill_params(oracle.TransformerDecoder(oracle.TransformerDecoderLayer(E, h, dim_feedforward=E, dropout=0.1, activation="gelu"), layers), 5).eval()
    tgt, mem = _rand(T, N, E, seed=1), _rand(S, N, E, seed=2)
    tpad, mpad = torch.zeros(N, T, dtype=torch.bool), torch.zeros(N, S, dtype=torch.bool)
    mpad[2, 15:] = True
    causal = torch.triu(torch.full((T, T), float("-inf")), 1)
    with torch.no_grad():
        y0, a0, b0 = dec(tgt, mem, tgt_mask=causal.to(DEV), tgt_key_padding_mask=tpad.to(DEV), memory_key_padding_mask=mpad.to(DEV))
        dec.return_attention = True
        y1, a1, b1 = dec(tgt, mem, tgt_mask=causal.to(DEV), tgt_key_padding_mask=tpad.to(DEV), memory_key_padding_mask=mpad.to(DEV))
        want = ref(tgt.cpu(), mem.cpu(), tgt_mask=causal, tgt_key_padding_mask=tpad, memory_key_padding_mask=mpad)
    assert a0 is None and b0 is None
    _close(y1.cpu(), y0.cpu(), 1e-5, "output with and without the weights")
    _close(y1.cpu(), want[0], 1e-4, "decoder output")
    _close(a1.cpu(), want[1], 1e-4, "self-attention weights of the last layer")
    _close(b1.cpu(), want[2], 1e-4, "memory-attention weights of the last layer")


def test_concatenation_backward_inside_the_layernorm_backward_matches_the_two_kernels():
    """case_layernorm_bwd_concat5 (ops.concat5_layer_norm_carry): Interaction -> first TransformerBlock at H = 512 in bf16 -- the gradients
    of the encodings and of every parameter equal those of the separate LayerNorm / concat5 backward kernels (dG rounded to bf16 in both),
    the fused kernel is the one launched, padded rows get zero gradients, forward values are identical."""
    import case_rg_amd
    from case_rg_amd import _abi
    from case_rg_amd.utils import fill_params
    ops = _ops()
    ns = case_rg_amd.namespace()
    case_rg_amd.set_compute_dtype(torch.bfloat16)
    try:
        H, B, P, Lp, Lq = 512, 2, 3, 72, 24
        inter = fill_params(ns.Interaction(H), 5).to(DEV)
        block = fill_params(ns.TransformerBlock(8, 5 * H, H), 6).to(DEV).train()
        eq0, ep0 = _rand(B, 1, Lq, H, dt=torch.bfloat16, seed=1), _rand(B, P, Lp, H, dt=torch.bfloat16, seed=2)
        qm = torch.ones(B, 1, Lq, dtype=torch.bool, device=DEV)
        pm = torch.ones(B, P, Lp, dtype=torch.bool, device=DEV)
        pm[0, 1, 50:] = False
        pm[1, 2, 1:] = False
        gout = _rand(B, P, Lp, H, dt=torch.bfloat16, seed=3)
        res = {}
        for fused in (True, False):
            ops.CONCAT5_LN = fused
            eq, ep = eq0.clone().requires_grad_(True), ep0.clone().requires_grad_(True)
            for p_ in list(inter.parameters()) + list(block.parameters()):
                p_.grad = None
            calls = {}
            raw = _abi.call

            def counting(name, *a):
                calls[name] = calls.get(name, 0) + 1
                return raw(name, *a)

            _abi.call = counting
            try:
                _, g_qp = inter(eq, ep, qm, pm)
                y = block(g_qp, pm)
                y.backward(gout)
            finally:
                _abi.call = raw
            assert calls.get("case_layernorm_bwd_concat5", 0) == (1 if fused else 0)
            # the passage side's concatenation backward is inside the fused kernel (the query side's output is unused in this test)
            assert calls.get("case_concat5_bwd", 0) == (0 if fused else 1)
            res[fused] = [y.detach(), eq.grad, ep.grad] + [p_.grad for p_ in list(inter.parameters()) + list(block.parameters())]
        assert torch.equal(res[True][0], res[False][0])
        for a, b in zip(res[True][1:], res[False][1:]):
            _close(a, b, 4e-3, "gradient through the fused concatenation backward")
    finally:
        ops.CONCAT5_LN = True
        case_rg_amd.set_compute_dtype(torch.float32)


@pytest.mark.parametrize("which", ["linear", "ffn"])
def test_layernorm_as_the_tail_of_linear_and_ffn_equals_the_separate_ops(which):
    """ops.linear(ln=...) / ops.ffn(ln=...): identical forward, and gradients equal to the composition with ops.layer_norm (the backward
    differs only in WHERE the dropout-masked gradient is produced), with dropout on and the residual forms the layers use."""
    from case_rg_amd import config
    ops = _ops()
    dt = torch.bfloat16
    M, K = 384, 512
    x0 = _rand(M, K, dt=dt, seed=1)
    w1, b1, w2, b2 = _rand(512, K, seed=2, scale=K ** -0.5), _rand(512, seed=3), _rand(512, 512, seed=4, scale=512 ** -0.5), _rand(512, seed=5)
    gam, bet = _rand(512, seed=6) + 1.0, _rand(512, seed=7)
    gout = _rand(M, 512, dt=dt, seed=8)
    config.set_dropout(True)
    try:
        res = {}
        for fused in (True, False):
            config.manual_seed(5)
            ops.LN_TAIL = fused
            leaves = [t.clone().requires_grad_(True) for t in (x0, w1, b1, w2, b2, gam, bet)]
            x, a1, c1, a2, c2, g, b = leaves
            if which == "linear":
                y = ops.linear(x, a1, c1, residual=x, p_drop=0.1, ln=(g, b, 1e-5))
            else:
                y = ops.ffn(x, a1, c1, a2, c2, "gelu", p_inner=0.1, p_out=0.1, residual=x, ln=(g, b, 1e-5))
            y.backward(gout)
            res[fused] = [y.detach()] + [t.grad for t in leaves if t.grad is not None]
        assert torch.equal(res[True][0], res[False][0]), "forward differs"
        assert len(res[True]) == len(res[False])
        for a, b in zip(res[True][1:], res[False][1:]):
            _close(a, b, 2e-3, "%s gradient with the LayerNorm tail" % which)
    finally:
        ops.LN_TAIL = True
        config.set_dropout(False)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_fanout_sums_the_gradients_of_its_aliases_in_one_launch(dt):
    """ops.fanout / case_add_n: n aliases for n consumers, ONE summing kernel in backward (f32 accumulation, one rounding); falls back to
    plain adds for what the kernel does not take (odd sizes), is the identity without gradients."""
    ops = _ops()
    from case_rg_amd import _abi
    for shape, n in (((6, 40, 64), 5), ((3, 7, 64), 2), ((5, 3), 3)):
        x = _rand(*shape, dt=dt, seed=1).requires_grad_(True)
        ws = [_rand(*shape, dt=dt, seed=10 + i) for i in range(n)]
        calls = {}
        raw = _abi.call

        def counting(name, *a):
            calls[name] = calls.get(name, 0) + 1
            return raw(name, *a)

        _abi.call = counting
        try:
            outs = ops.fanout(x, n)
            assert len(outs) == n and all(o.data_ptr() == x.data_ptr() for o in outs)
            loss = sum((o.float() * w.float()).sum() for o, w in zip(outs, ws))
            loss.backward()
        finally:
            _abi.call = raw
        want = sum(w.float() for w in ws)
        _close(x.grad, want, 1e-6 if dt == torch.float32 else 8e-3, "fan-out gradient %s x %d" % (shape, n))
        fits = x.numel() % (8 if dt == torch.bfloat16 else 4) == 0
        assert calls.get("case_add_n", 0) == (1 if fits else 0)
    with torch.no_grad():
        y = _rand(4, 64, dt=dt, seed=2)
        assert all(o is y for o in ops.fanout(y, 3))


def test_zero_arena_serves_one_fill_per_step_with_disjoint_zeroed_slices():
    """ops._ZeroArena: from the second step on (steps = optimizer updates = ops.PARAM_EPOCH) every zero-initialised gradient buffer of
    the backward pass is a slice of ONE zero-filled allocation sized by the previous step; slices are zero when handed out, disjoint,
    256-byte aligned, and requests beyond the estimate fall back to their own allocation."""
    ops = _ops()
    dev = torch.device("cuda", torch.cuda.current_device())
    shapes = [(512, 512), (512,), (7,), (1536, 512), (1,)]
    ops._ZEROS.__init__()  # forget what earlier tests of this process took
    storages = []
    for step in range(3):
        ops.invalidate_param_cache()  # what the optimizer does after writing the parameters
        got = []
        for rep in range(2):
            got += ops._zeros_like_shapes(dev, *shapes)
        extra = ops._zeros_like_shapes(dev, (300, 300))[0] if step == 2 else None  # beyond what step 1 took
        for t, sh in zip(got, shapes * 2):
            assert t.shape == torch.Size(sh) and t.dtype == torch.float32 and not t.any().item()
        for i, t in enumerate(got):
            t.fill_(float(i + 1))
        for i, t in enumerate(got):
            assert (t == float(i + 1)).all().item(), "slices overlap"
        storages.append({t.untyped_storage().data_ptr() for t in got})
        if extra is not None:
            assert not extra.any().item() and extra.untyped_storage().data_ptr() not in storages[-1]
        del got
    assert len(storages[0]) == 2, "first step: no estimate yet, one allocation per request"
    assert len(storages[1]) == 1 and len(storages[2]) == 1, "later steps: one arena"


def test_weight_gradient_through_split_slabs_is_exact_and_repeatable():
    """case_gemm_dw_slabs: the deeply split weight gradients (ops.gemm sends split_k >= DW_SLAB_MIN_SPLIT here) store one f32 slab per
    split and sum them in a fixed order -- same value as the atomic form to f32 rounding, bit-identical from launch to launch, C is
    accumulated into (+=), and the fused bias gradient still arrives."""
    ops = _ops()
    from case_rg_amd import _abi as A
    assert A.lib.case_abi_features() & A.FEAT_GEMM_DW_SLABS
    dt = torch.bfloat16
    # the last two shapes give every persistent workgroup SEVERAL output tiles (45 tiles x 17 splits = 765, 30 x 23 = 690 on 256 CUs): the
    # wait in front of a workgroup's 2nd .. nth tile counts the previous epilogue's stores (round 4 counted too many for the slab
    # epilogue -- ADVICE r4 -- and the fragment reads could overtake the first DMAs; one tile per workgroup never reaches that wait)
    for (Mtok, N, K, split, bias) in ((64 * 64, 512, 512, 16, True), (64 * 96, 256, 768, 12, False), (64 * 27, 512, 256, 9, True),
                                      (64 * 17 * 3, 768, 3840, 17, True), (64 * 23, 1536, 1280, 23, False)):
        g, x = _rand(Mtok, N, dt=dt, seed=1), _rand(Mtok, K, dt=dt, seed=2, scale=Mtok ** -0.5)
        ref = g.float().t() @ x.float()
        d = A.GemmDesc()
        d.M, d.N, d.K, d.lda, d.ldb, d.ldc, d.split_k = N, K, Mtok, N, K, K, split
        assert A.lib.case_gemm_dw_slab_bytes(d) == split * N * K * 4 == A.lib.case_workspace_bytes(A.WS_GEMM_DW_SLABS, ctypes.addressof(d), 0)
        outs = []
        for rep in range(3):
            dw = torch.full((N, K), 0.5 if rep == 2 else 0.0, device="cuda", dtype=torch.float32)
            db = torch.zeros(N, device="cuda", dtype=torch.float32) if bias else None
            trace = ops.TILE_TRACE = []
            ops.GEMM_TILE = 256  # (the cost model would give test-sized problems to the 128 tiling)
            try:
                ops.gemm(g, x, dw, N, K, Mtok, N, K, K, a_kmajor=True, b_kmajor=True, split_k=split, epilogue=A.EPI_ATOMIC, rowsum_out=db)
            finally:
                ops.GEMM_TILE, ops.TILE_TRACE = 0, None
            assert trace[0] == 256
            torch.cuda.synchronize()
            outs.append(dw)
            if bias:
                _close(db, g.float().sum(0), 2e-3, "bias gradient beside the slabs N=%d" % N)
        _close(outs[0], ref, 2e-3, "slab dW %dx%d split %d" % (N, K, split))
        assert torch.equal(outs[0], outs[1]), "slab weight gradient differs between two launches"
        _close(outs[2], ref + 0.5, 2e-3, "slab dW accumulates into C")
    # the entry point refuses what it cannot run, and a short workspace
    d = A.GemmDesc()
    d.M, d.N, d.K, d.lda, d.ldb, d.ldc, d.split_k = 512, 512, 4096, 512, 512, 512, 16
    d.batch1 = d.batch2 = 1
    d.a_kmajor = d.b_kmajor = 1
    d.in_dtype, d.out_dtype, d.epilogue, d.alpha, d.tile = A.BF16, A.F32, A.EPI_ATOMIC, 1.0, 256
    g, x = _rand(4096, 512, dt=dt, seed=1), _rand(4096, 512, dt=dt, seed=2)
    dw = torch.zeros(512, 512, device="cuda")
    ws = torch.empty(1024, dtype=torch.uint8, device="cuda")
    with pytest.raises(RuntimeError, match="workspace"):
        A.call("case_gemm_dw_slabs", d, g.data_ptr(), x.data_ptr(), dw.data_ptr(), None, ws.data_ptr(), ws.numel(), 0)
    d.split_k = 1
    with pytest.raises(RuntimeError, match="split_k"):
        A.call("case_gemm_dw_slabs", d, g.data_ptr(), x.data_ptr(), dw.data_ptr(), None, ws.data_ptr(), ws.numel(), 0)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("T", [1, 2])
def test_additive_scores_decode_rows(dt, T):
    """T <= 2 takes the row-wise (decode-step) kernel."""
    ops = _ops()
    B, S, H = 3, 77, 128
    wq, uh, v = _rand(B, T, H, seed=1), _rand(B, S, H, dt=dt, seed=2), _rand(H, seed=3)
    s = ops.additive_scores(wq, uh, v)
    ref = torch.tanh(wq[:, :, None, :] + uh.float()[:, None, :, :]) @ v
    _close(s, ref, 1e-3 if dt == torch.float32 else 1e-2, "additive decode rows")


def test_cast_cache_is_not_fooled_by_address_reuse():
    """bf16 operand copies are cached for Parameters only; a temporary weight at a recycled address must not hit the cache."""
    ops = _ops()
    x = _rand(64, 64, dt=torch.bfloat16, seed=1)
    outs = []
    for seed in (2, 3, 4):
        w = _rand(64, 64, seed=seed)  # freed at the end of the iteration: the allocator hands the address out again
        outs.append((ops.linear(x, w, None).float(), F.linear(x.float(), w.to(torch.bfloat16).float())))
        del w
    for got, want in outs:
        _close(got, want, 3e-2, "temporary weight")
    p = torch.nn.Parameter(_rand(64, 64, seed=9))
    a = ops.linear(x, p, None).float()
    with torch.no_grad():
        p.mul_(2.0)  # in-place update bumps the version: the cached copy must be refreshed
    b = ops.linear(x, p, None).float()
    _close(b, 2 * a, 3e-2, "parameter update invalidates the cached copy")


def test_bf16_parameter_cache_follows_data_swaps_and_explicit_invalidation():
    """cast_param caches the bf16 copy of an f32 Parameter.  ``p.data = t`` (EMA.apply_shadow / restore) changes data_ptr and
    must be seen; in-place writes through ``p.data`` move neither ``_version`` nor the address, so the code paths that do that
    (init_params, GradSync.broadcast_parameters, the trainer after optimizer.step()) call invalidate_param_cache()."""
    ops = _ops()
    from case_rg_amd.common.EMA import EMA
    lin = torch.nn.Linear(64, 32, bias=False).to(DEV)
    x = _rand(8, 64, dt=torch.bfloat16, seed=1)
    y1 = ops.linear(x, lin.weight).float()
    v0 = lin.weight._version
    lin.weight.data = lin.weight.data * 2.0                      # storage swap
    assert lin.weight._version == v0
    _close(ops.linear(x, lin.weight), 2.0 * y1, 1e-2, "after p.data = t")
    ema = EMA(lin, 0.5)
    ema.register()                                               # shadow = 2 w
    with torch.no_grad():
        lin.weight.mul_(2.0)                                     # live = 4 w (in place: bumps _version)
    _close(ops.linear(x, lin.weight), 4.0 * y1, 1e-2, "after an in-place update")
    ema.apply_shadow()
    _close(ops.linear(x, lin.weight), 2.0 * y1, 1e-2, "EMA weights applied")
    ema.restore()
    _close(ops.linear(x, lin.weight), 4.0 * y1, 1e-2, "EMA weights restored")
    lin.weight.data.mul_(0.25)                                   # invisible to _version and data_ptr
    ops.invalidate_param_cache()
    _close(ops.linear(x, lin.weight), y1, 1e-2, "after invalidate_param_cache()")
    del lin, ema
    import gc
    gc.collect()
    live = torch.nn.Parameter(torch.zeros(4, 4, device=DEV))
    ops.cast_param(live, torch.bfloat16)  # a miss evicts the entries of dead owners
    assert all(entry[0]() is not None for entry in ops._cast_cache.values())


def test_gemm_rejects_mismatched_aux_dtype():
    ops = _ops()
    x = _rand(16, 32, dt=torch.bfloat16, seed=1)
    w = _rand(8, 32, seed=2)
    with pytest.raises(TypeError, match="aux"):
        ops.linear(x, w, residual=_rand(16, 8, seed=3))          # f32 residual next to bf16 activations


@pytest.mark.parametrize("M,N,act,with_res", [(256, 1536, None, False), (64, 512, "gelu", False), (320, 4096, None, False), (128, 512, "relu", False),
                                              (256, 512, None, True)])
def test_layernorm_prologue_gemm_matches_layernorm_then_linear(M, N, act, with_res):
    """case_gemm_ln (the greedy step's LN -> projection pairs in one launch): LN(x) and act(LN(x) W^T + b) (+ residual) against
    ops.layer_norm + ops.linear / ops.ffn's first half on the same bf16 rows, and against f32 torch."""
    ops = _ops()
    dt = torch.bfloat16
    old_mode, ops.LN_GEMM = ops.LN_GEMM, "on"  # (measured flat on the greedy step, so not the default: ops.LN_GEMM)
    try:
        _ln_prologue_case(ops, M, N, act, with_res, dt)
    finally:
        ops.LN_GEMM = old_mode


def _ln_prologue_case(ops, M, N, act, with_res, dt):
    x = (_rand(M, 512, dt=dt, seed=1) * 1.7 + 0.3).to(dt)
    w, b = _rand(N, 512, seed=2, scale=512 ** -0.5), _rand(N, seed=3, scale=0.2)
    gamma, beta = 1.0 + _rand(512, seed=4, scale=0.1), _rand(512, seed=5, scale=0.1)
    res = _rand(M, N, dt=dt, seed=6) if with_res else None
    with torch.no_grad():
        assert ops.ln_gemm_supported(x, N)
        y, xn = ops.ln_linear(x, (gamma, beta, 1e-5), w, b, act=act, residual=res)
        xn0 = ops.layer_norm(x, gamma, beta, 1e-5)
    # LN rows: the same two-pass f32 statistics as the stand-alone kernel -> at most one bf16 ulp apart on a few elements
    assert (xn.float() - xn0.float()).abs().max().item() <= 2 ** -7 * xn0.float().abs().max().item()
    assert (xn != xn0).float().mean().item() < 0.01
    ref = F.linear(F.layer_norm(x.float(), (512,), gamma, beta, 1e-5).to(dt).float(), w.to(dt).float(), b)
    ref = F.gelu(ref) if act == "gelu" else torch.relu(ref) if act == "relu" else ref
    if with_res:
        ref = ref + res.float()
    _close(y, ref, 2e-2, "ln_linear N=%d act=%s" % (N, act))
    # where the fused launch does not apply (33 rows) the same call falls back to two launches with the same values
    with torch.no_grad():
        xs = x[:33].contiguous()
        assert not ops.ln_gemm_supported(xs, N)
        if act is None:
            y2, xn2 = ops.ln_linear(xs, (gamma, beta, 1e-5), w, b, residual=None if res is None else res[:33].contiguous())
            _close(y2, ref[:33], 2e-2, "ln_linear fallback")
            assert torch.equal(xn2, xn0[:33])


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_multihead_attention_takes_an_arbitrary_attn_mask(dt):
    """nn.MultiheadAttention's attn_mask beyond the causal pattern (VERDICT r4 missing 7; no caller on the CaSE / Masque path): a float
    additive mask and a bool mask (True = masked) against torch's own module with the same weights, forward and backward."""
    import case_rg_amd
    from case_rg_amd.common.attention import MultiheadAttention
    case_rg_amd.set_compute_dtype(dt)
    case_rg_amd.set_dropout(False)
    try:
        torch.manual_seed(9)
        E, h, Lq, Lk, N = 64, 4, 10, 13, 3
        ours = MultiheadAttention(E, h).to(DEV)
        ref = torch.nn.MultiheadAttention(E, h).to(DEV)
        with torch.no_grad():
            ours.in_proj_bias.copy_(torch.randn(3 * E) * 0.1)
            ref.in_proj_weight.copy_(ours.in_proj_weight.to(dt).float()); ref.in_proj_bias.copy_(ours.in_proj_bias)
            ref.out_proj.weight.copy_(ours.out_proj.weight.to(dt).float()); ref.out_proj.bias.copy_(ours.out_proj.bias)
        case_rg_amd.ops.invalidate_param_cache()
        g = torch.Generator().manual_seed(4)
        fmask = (torch.randn(Lq, Lk, generator=g) * 2).to(DEV)
        bmask = (torch.rand(Lq, Lk, generator=g) < 0.3).to(DEV)
        bmask[:, 0] = False  # no fully masked row (torch gives NaN there, so do we)
        pad = torch.zeros(N, Lk, dtype=torch.bool, device=DEV)
        pad[1, 9:] = True
        tol = 1e-3 if dt == torch.float32 else 3e-2
        for mask in (fmask, bmask):
            q = _rand(Lq, N, E, dt=dt, seed=1).requires_grad_()
            kv = _rand(Lk, N, E, dt=dt, seed=2).requires_grad_()
            out, _ = ours(q, kv, kv, attn_mask=mask, key_padding_mask=pad)
            qr, kr = q.detach().float().requires_grad_(), kv.detach().float().requires_grad_()
            want, _ = ref(qr, kr, kr, attn_mask=mask, key_padding_mask=pad, need_weights=False)
            _close(out, want, tol, "attn_mask forward")
            gr = _rand(Lq, N, E, dt=dt, seed=3)
            out.backward(gr)
            want.backward(gr.float())
            _close(q.grad, qr.grad, 2 * tol, "attn_mask dq")
            _close(kv.grad, kr.grad, 2 * tol, "attn_mask dkv")
        # round 6: torch's per-head 3-D form [N * heads, Lq, Lk], and need_weights together with a mask (head-averaged probabilities)
        m3 = (torch.randn(N * h, Lq, Lk, generator=g) * 2).to(DEV)
        q = _rand(Lq, N, E, dt=dt, seed=1).requires_grad_()
        kv = _rand(Lk, N, E, dt=dt, seed=2).requires_grad_()
        out, w = ours(q, kv, kv, attn_mask=m3, key_padding_mask=pad, need_weights=True)
        qr, kr = q.detach().float().requires_grad_(), kv.detach().float().requires_grad_()
        want, w_ref = ref(qr, kr, kr, attn_mask=m3, key_padding_mask=pad, need_weights=True)
        _close(out, want, tol, "3-D attn_mask forward")
        _close(w, w_ref, tol, "head-averaged weights under a mask")
        gr = _rand(Lq, N, E, dt=dt, seed=3)
        out.backward(gr)
        want.backward(gr.float())
        _close(q.grad, qr.grad, 2 * tol, "3-D attn_mask dq")
        _close(kv.grad, kr.grad, 2 * tol, "3-D attn_mask dkv")
        # the causal pattern given as a plain float matrix is still recognised and takes the fused path's flag
        from case_rg_amd.common.attention import split_attn_mask
        n = 6
        cm = torch.triu(torch.full((n, n), -1e20, device=DEV), 1)
        assert split_attn_mask(cm) == (True, None)
        assert split_attn_mask(None) == (False, None)
    finally:
        case_rg_amd.set_compute_dtype(torch.float32)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cols", [512, 100])
def test_mask_rows_in_place_touches_only_the_invalid_rows(dt, cols):
    """ops.mask_rows(in_place=True) without autograd zeroes the padded rows IN the tensor (case_mask_rows with y == x: the valid rows are
    neither read nor written); with autograd it stays the out-of-place differentiable op (common/TransformerBlock.py:31 semantics)."""
    ops = _ops()
    x = _rand(37, cols, dt=dt, seed=1)
    valid = torch.rand(37, device=DEV) > 0.4
    want = x * valid[:, None].to(x.dtype)
    with torch.no_grad():
        y = ops.mask_rows(x.clone(), valid)
        z0 = x.clone()
        z = ops.mask_rows(z0, valid, in_place=True)
    assert torch.equal(y, want) and z.data_ptr() == z0.data_ptr() and torch.equal(z, want)
    xg = x.clone().requires_grad_()
    out = ops.mask_rows(xg * 1.0, valid, in_place=True)
    out.sum().backward()
    assert torch.equal(out, want) and torch.equal(xg.grad, valid[:, None].to(x.dtype).expand_as(x))


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_layers_take_arbitrary_src_tgt_and_memory_masks(dt):
    """VERDICT r5 missing 7: the reference's layers hand ``src_mask`` / ``tgt_mask`` / ``memory_mask`` to nn.MultiheadAttention
    (common/TransformerEncoder.py:66-68, common/TransformerDecoder.py:76-82); beyond the causal pattern these used to raise here.  A band
    mask on the encoder, a bool tgt_mask and a float memory_mask on the decoder, layer and stack level, against the CPU oracle's layers
    (which restate the reference's) with the same weights; forward and the input gradients.  No caller on the CaSE / Masque path passes one."""
    import case_rg_amd
    import oracle
    from case_rg_amd.utils import fill_params
    case_rg_amd.set_compute_dtype(dt)
    case_rg_amd.set_dropout(False)
    try:
        ns = case_rg_amd.namespace()
        E, h, L, S, N = 64, 4, 12, 9, 3
        g = torch.Generator().manual_seed(21)
        band = torch.full((L, L), float("-inf"))
        for i in range(L):
            band[i, max(0, i - 2):i + 3] = 0.0                      # every row keeps its neighbourhood
        tmask = torch.rand(L, L, generator=g) < 0.3
        tmask[torch.arange(L), torch.arange(L)] = False               # bool, True = masked
        tmask[:, 1] = False                                           # ... and no fully masked row once key 0 of sequence 1 is padding
        mmask = torch.randn(L, S, generator=g) * 1.5                  # float additive memory mask
        pad = torch.zeros(N, L, dtype=torch.bool)
        pad[1, :1] = True                                             # (a padded key inside every band it touches would be fine; a row whose
        mpad = torch.zeros(N, S, dtype=torch.bool)                    #  whole band is padding gives NaN, here as in torch)
        mpad[2, 6:] = True
        src = torch.randn(L, N, E, generator=g)
        mem = torch.randn(S, N, E, generator=g)
        tol = 2e-3 if dt == torch.float32 else 5e-2

        def both(make_ours, make_ref, run):
            ours = fill_params(make_ours(), 5).to(DEV).train()
            ref = fill_params(make_ref(), 5).train()
            case_rg_amd.ops.invalidate_param_cache()
            x = src.to(DEV).to(dt).requires_grad_()
            xr = x.detach().float().cpu().requires_grad_()
            y, yr = run(ours, x, DEV, dt), run(ref, xr, "cpu", torch.float32)
            _close(y, yr.to(DEV), tol, "masked layer forward")
            go = torch.randn(y.shape, generator=torch.Generator().manual_seed(3))
            y.backward(go.to(DEV).to(y.dtype))
            yr.backward(go)
            _close(x.grad, xr.grad.to(DEV), 3 * tol, "masked layer dx")

        enc = lambda m, x, dev, d: m(x, src_mask=band.to(dev), src_key_padding_mask=pad.to(dev))
        both(lambda: ns.TransformerEncoderLayer(E, h, dim_feedforward=E, dropout=0.1, activation="gelu"),
             lambda: oracle.TransformerEncoderLayer(E, h, dim_feedforward=E, dropout=0.0, activation="gelu"), enc)
        both(lambda: ns.TransformerEncoder(ns.TransformerEncoderLayer(E, h, dim_feedforward=E, dropout=0.1, activation="gelu"), 2),
             lambda: oracle.TransformerEncoder(oracle.TransformerEncoderLayer(E, h, dim_feedforward=E, dropout=0.0, activation="gelu"), 2),
             lambda m, x, dev, d: m(x, mask=band.to(dev), src_key_padding_mask=pad.to(dev)))
        tmask_f = torch.zeros(L, L).masked_fill(tmask, float("-inf"))  # (the oracle's attention adds its mask: the bool mask in float form)
        dec = lambda m, x, dev, d: m(x, mem.to(dev).to(d), tgt_mask=tmask.to(dev) if dev != "cpu" else tmask_f, memory_mask=mmask.to(dev),
                                     tgt_key_padding_mask=pad.to(dev), memory_key_padding_mask=mpad.to(dev))[0]
        both(lambda: ns.TransformerDecoderLayer(E, h, dim_feedforward=E, dropout=0.1, activation="gelu"),
             lambda: oracle.TransformerDecoderLayer(E, h, dim_feedforward=E, dropout=0.0, activation="gelu"), dec)
        both(lambda: ns.TransformerDecoder(ns.TransformerDecoderLayer(E, h, dim_feedforward=E, dropout=0.1, activation="gelu"), 2),
             lambda: oracle.TransformerDecoder(oracle.TransformerDecoderLayer(E, h, dim_feedforward=E, dropout=0.0, activation="gelu"), 2), dec)
    finally:
        case_rg_amd.set_compute_dtype(torch.float32)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_bilinear_attention_general_masks_and_leading_dims(dt):
    """common/BilinearAttention.py:13-59 beyond the path's call forms (VERDICT r5 missing 7): a mask that is NOT an outer product of row and
    column validity, 4-D inputs [B, n, T, *] with a 4-D mask, and the softmax over the QUERY axis -- forward (context, raw scores,
    probabilities) and the gradients of query / key / value against the CPU oracle's module with the same weights."""
    import case_rg_amd
    import oracle
    from case_rg_amd.utils import fill_params
    case_rg_amd.set_compute_dtype(dt)
    case_rg_amd.set_dropout(False)
    try:
        ns = case_rg_amd.namespace()
        B, n, T, S, Q, K, Hh = 2, 3, 5, 11, 24, 16, 32
        g = torch.Generator().manual_seed(17)
        ours = fill_params(ns.BilinearAttention(Q, K, Hh), 7).to(DEV)
        ref = fill_params(oracle.BilinearAttention(Q, K, Hh), 7)
        case_rg_amd.ops.invalidate_param_cache()
        tol = 2e-3 if dt == torch.float32 else 4e-2
        for shape4 in (False, True):
            lead = (B, n) if shape4 else (B,)
            q0, k0, v0 = torch.randn(*lead, T, Q, generator=g), torch.randn(*lead, S, K, generator=g), torch.randn(*lead, S, K, generator=g)
            mask = torch.rand(*lead, T, S, generator=g) < 0.7   # a general mask: not row x column
            mask[..., 0] = True                                  # (a row without any admissible key: zeros on both sides, tested below)
            mask[(0,) * len(lead) + (2,)] = False
            q, k, v = [t.to(DEV).to(dt).requires_grad_() for t in (q0, k0, v0)]
            qr, kr, vr = [t.detach().float().cpu().requires_grad_() for t in (q, k, v)]
            ctx, s, p = ours(q, k, v, mask=mask.to(DEV))
            ctx_r, s_r, p_r = ref(qr, kr, vr, mask=mask)
            p_r = p_r.nan_to_num(0.0)  # (torch: softmax of an all -inf row is NaN, then masked_fill(~mask, 0) -> 0)
            assert ctx.shape == ctx_r.shape and p.shape == p_r.shape
            _close(p, p_r.to(DEV), tol, "general-mask probabilities")
            assert (p.float()[~mask.to(DEV)] == 0).all() and (p.float()[(0,) * len(lead) + (2,)] == 0).all()
            fin = torch.isfinite(s_r)
            assert torch.equal(torch.isfinite(s.float()).cpu(), fin)
            _close(s.float().cpu()[fin], s_r[fin], tol, "raw scores")
            ctx_r2 = (p_r.reshape(-1, T, S) @ vr.reshape(-1, S, K)).reshape(ctx_r.shape)  # (the oracle's ctx carries the NaN row)
            _close(ctx, ctx_r2.to(DEV), tol, "context")
            go = torch.randn(ctx.shape, generator=g)
            ctx.backward(go.to(DEV).to(ctx.dtype))
            ctx_r2.backward(go)
            for name, a, b in (("dq", q.grad, qr.grad), ("dk", k.grad, kr.grad), ("dv", v.grad, vr.grad)):
                _close(a, b.to(DEV), 3 * tol, "general-mask " + name)
        # softmax over the query axis (softmax_dim = -2), outer-product mask
        q, k = torch.randn(B, T, Q, generator=g).to(DEV).to(dt), torch.randn(B, S, K, generator=g).to(DEV).to(dt)
        rv, cv = torch.rand(B, T, generator=g) < 0.8, torch.rand(B, S, generator=g) < 0.8
        rv[:, 0], cv[:, 0] = True, True
        mask = rv[:, :, None] & cv[:, None, :]
        _, p = ours.score(q, k, softmax_dim=-2, mask=mask.to(DEV))
        _, p_r = ref.score(q.float().cpu(), k.float().cpu(), softmax_dim=-2, mask=mask)
        _close(p, p_r.nan_to_num(0.0).to(DEV), tol, "softmax over the query axis")
    finally:
        case_rg_amd.set_compute_dtype(torch.float32)
