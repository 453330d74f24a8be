"""K22 (csrc/attn_pointer.hip): the greedy step's additive attention -- scores over the cached e^{2 uh} rows, masked softmax, the copy prior's
renormalisation and the context product in one launch (common/BilinearAttention.py:31-59 at T = 1, CaSE/Model.py:79-82)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rand(*shape, seed=0, scale=1.0, dt=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV).to(dt)


@pytest.mark.parametrize("B,S,with_prior", [(5, 3840, True), (3, 64, False), (130, 333, True), (2, 1, True)])
def test_pointer_attend_decode_matches_f32_restatement(B, S, with_prior):
    from case_rg_amd import ops
    H = 512
    wq = _rand(B, H, seed=1, scale=1.5)
    uh = _rand(B, S, H, seed=2, scale=1.5)
    uh[0, 0, :7] = torch.tensor([30.0, -30.0, 25.0, -25.0, 50.0, -50.0, 0.0], device=DEV)  # saturated and clamped arguments
    wq[0, :7] = torch.tensor([-30.0, 30.0, 25.0, -25.0, 3.0, -3.0, 45.0], device=DEV)
    v = _rand(H, seed=3, scale=0.3)
    mem = _rand(B, S, H, seed=4, dt=torch.bfloat16)
    g = torch.Generator().manual_seed(5)
    cv = (torch.rand(B, S, generator=g) < 0.85).to(DEV)
    cv[0, 0] = True
    rv = torch.ones(B, dtype=torch.bool, device=DEV)
    if B > 2:
        cv[1] = False   # no valid source position
        rv[2] = False   # PAD target row
    prior = torch.rand(B, S, generator=g).to(DEV) if with_prior else None
    eu = ops.additive_key_exp(uh)
    ctx, p, copy = ops.pointer_attend_decode(wq, eu, v, mem, cv, rv, prior)
    # f32 restatement (tanh of the unclamped sum; the seven planted features cover +-60, cancelling +-30 and the clamp at 43)
    s = (torch.tanh(wq[:, None, :] + uh) * v).sum(-1).masked_fill(~cv, float("-inf"))
    pr = torch.nan_to_num(torch.softmax(s, -1), nan=0.0) * rv[:, None]
    cr = torch.einsum("bs,bsh->bh", pr, mem.float())
    # e^{2 uh} is stored in bf16: a relative error of 2^-9 on the exponential moves a tanh by <= 2^-10, the score (512 terms, |v| ~ 0.3) by ~1e-2
    assert (p - pr).abs().max().item() <= 3e-2 * pr.abs().max().clamp_min(1e-6).item() + 2e-3
    assert (ctx.float() - cr).abs().max().item() <= 3e-2 * cr.abs().max().clamp_min(1e-3).item()
    assert torch.allclose(p.sum(-1), (rv & cv.any(-1)).float(), atol=1e-5)
    if B > 2:
        assert float(p[1].abs().max()) == 0.0 and float(ctx[1].float().abs().max()) == 0.0
        assert float(p[2].abs().max()) == 0.0 and float(ctx[2].float().abs().max()) == 0.0
    assert float(p.masked_select(~cv).abs().max()) == 0.0 if (~cv).any() else True
    if with_prior:
        want = p * prior
        want = want / (1e-8 + want.sum(-1, keepdim=True))
        assert torch.allclose(copy, want, rtol=2e-5, atol=1e-7)
    else:
        assert copy is None
    # fixed-order sums: bit-identical from launch to launch
    ctx2, p2, _ = ops.pointer_attend_decode(wq, eu, v, mem, cv, rv, prior)
    assert torch.equal(ctx, ctx2) and torch.equal(p, p2)
    # wq_add: the step-invariant part of the query projection is added inside the kernel, in f32 -- the same bits as the sum handed over whole
    part = _rand(B, H, seed=6, scale=0.8)
    ctx3, p3, copy3 = ops.pointer_attend_decode(wq - part, eu, v, mem, cv, rv, prior, wq_add=part)
    ctx4, p4, copy4 = ops.pointer_attend_decode((wq - part) + part, eu, v, mem, cv, rv, prior)
    assert torch.equal(ctx3, ctx4) and torch.equal(p3, p4) and (copy3 is None or torch.equal(copy3, copy4))


def test_split_query_equals_the_concatenated_query():
    """BilinearAttention.split_query + attend_decode(split=) (a step projects x_t alone, the feature half of the query projection is added in K22)
    against attend_decode on the concatenated [x_t | feature] query: same module, same bf16 inputs; the two differ in f32 summation order only."""
    from case_rg_amd import ops
    from case_rg_amd.common.BilinearAttention import BilinearAttention
    B, S, H = 64, 640, 512
    m = BilinearAttention(2 * H, H, H).to(DEV)
    g = torch.Generator().manual_seed(11)
    with torch.no_grad():
        for prm in m.parameters():
            prm.copy_((torch.randn(prm.shape, generator=g) * 0.05).to(DEV))
    x = _rand(B, 1, H, seed=12, dt=torch.bfloat16)
    feat = _rand(B, 1, H, seed=13, dt=torch.bfloat16)
    mem = _rand(B, S, H, seed=14, dt=torch.bfloat16)
    cv = (torch.rand(B, S, generator=g) < 0.9).to(DEV)
    cv[:, 0] = True
    rv = torch.ones(B, 1, dtype=torch.bool, device=DEV)
    prior = torch.rand(B, S, generator=g).to(DEV)
    with torch.no_grad():
        eu = m.project_keys_exp(mem)
        c1, p1 = m.attend_decode(torch.cat([x, feat], dim=-1), mem, rv, cv, eu, prior)
        split = m.split_query(feat, H)
        c2, p2 = m.attend_decode(x, mem, rv, cv, eu, prior, split=split)
    assert split[0].shape == (H, H) and split[1].shape == (B, H) and split[1].dtype == torch.float32
    assert (p1 - p2).abs().max().item() <= 1e-4 * p1.abs().max().item() + 1e-7
    assert (c1.float() - c2.float()).abs().max().item() <= 1e-2 * c1.float().abs().max().item()


def test_attend_decode_agrees_with_the_four_launch_form():
    """BilinearAttention.attend_decode (K22) against attend (scores -> masked softmax -> cast -> product) + the prior arithmetic of the greedy
    loop, same module, same bf16 inputs."""
    import case_rg_amd
    from case_rg_amd.common.BilinearAttention import BilinearAttention
    case_rg_amd.set_compute_dtype(torch.bfloat16)
    case_rg_amd.set_dropout(False)
    try:
        torch.manual_seed(3)
        B, S, H = 4, 900, 512
        att = BilinearAttention(2 * H, H, H).to(DEV).eval()
        q = _rand(B, 1, 2 * H, seed=7, dt=torch.bfloat16)
        mem = _rand(B, S, H, seed=8, dt=torch.bfloat16)
        cv = torch.ones(B, S, dtype=torch.bool, device=DEV)
        cv[1, 500:] = False
        rv = torch.ones(B, 1, dtype=torch.bool, device=DEV)
        w = torch.rand(B, S, device=DEV)
        with torch.no_grad():
            ctx0, p0 = att.attend(q, mem, mem, row_valid=rv, col_valid=cv, uh=att.project_keys(mem))
            c0 = w.unsqueeze(1) * p0
            c0 = c0 / (1e-8 + c0.sum(-1, keepdim=True))
            ctx1, c1 = att.attend_decode(q, mem, rv, cv, att.project_keys_exp(mem), w)
        assert (ctx1.float() - ctx0.float()).abs().max().item() <= 3e-2 * ctx0.float().abs().max().item()
        assert (c1 - c0).abs().max().item() <= 3e-2 * c0.abs().max().item()
    finally:
        case_rg_amd.set_compute_dtype(torch.float32)


@pytest.mark.parametrize("B,V,lens", [(6, 30522, (64, 3840)), (3, 150, (5, 9)), (130, 1000, (40,)), (2, 33000, (7, 8, 9, 10))])
def test_pointer_head_decode_matches_the_separate_launches(B, V, lens):
    """K23 against softmax -> p0 x gen -> sorted scatter -> add -> argmax (the launches it replaces), same inputs: gen and dist to f32
    rounding, the ids exactly (incl. a planted tie: the lowest index wins)."""
    from case_rg_amd import ops
    g = torch.Generator().manual_seed(V)
    logits = (torch.randn(B, V, generator=g) * 3).to(DEV)
    mix = torch.randn(B, 1 + len(lens), generator=g).to(DEV)
    S = sum(lens)
    src = torch.randint(0, V, (B, S), generator=g).to(DEV)
    src[0, : S // 2] = src[0, 0]          # a long run of one token
    src[1 % B, -1] = V + 5                # an out-of-vocabulary id: no mass
    copies = [torch.rand(B, n, generator=g).to(DEV) for n in lens]
    copies[0][0, 0] = 0.0
    sm = ops.SortedSource(src, V)
    gen, dist, ids = ops.pointer_head_decode(logits, mix, sm, copies)
    gen0 = ops.masked_softmax(logits.view(B, 1, V))
    pm = torch.softmax(mix, -1)
    ptr = torch.cat([pm[:, k + 1:k + 2] * c for k, c in enumerate(copies)], dim=-1).view(B, 1, S)
    dist0 = pm[:, 0:1].unsqueeze(-1) * gen0 + ops.copy_scatter(sm, ptr, V)
    assert torch.allclose(gen, gen0.view(B, V), rtol=2e-5, atol=1e-9)
    assert torch.allclose(dist, dist0.view(B, V), rtol=3e-5, atol=1e-8)
    assert torch.equal(ids, ops.row_argmax(dist)[0]), "argmax of the fused head's own distribution"
    top2 = dist0.view(B, V).topk(2, dim=-1).values
    decisive = (top2[:, 0] - top2[:, 1]) > 1e-6 * top2[:, 0]
    assert torch.equal(ids[decisive], dist0.view(B, V).argmax(-1)[decisive])
    # a tie: two equal logits far above the rest, no pointer mass on either -> the lower index
    logits2 = torch.full((B, V), -5.0, device=DEV)
    logits2[:, 11] = 9.0
    logits2[:, 7] = 9.0
    src2 = torch.full((B, S), 3, device=DEV)
    _, dist2, ids2 = ops.pointer_head_decode(logits2, mix, ops.SortedSource(src2, V), copies, want_gen=False)
    assert ids2.tolist() == [3 if float(dist2[b, 3]) > float(dist2[b, 7]) else 7 for b in range(B)]
    # bit-identical from launch to launch
    gen3, dist3, ids3 = ops.pointer_head_decode(logits, mix, sm, copies)
    assert torch.equal(dist, dist3) and torch.equal(ids, ids3) and torch.equal(gen, gen3)
