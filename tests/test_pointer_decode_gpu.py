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
