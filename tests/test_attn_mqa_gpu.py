"""K21 (csrc/attn_mqa.hip): the decode step's cross-attention on the RAW memory rows with the K / V projections absorbed
(common/TransformerDecoder.py:81-82 at one position per sequence, CaSE/Model.py:94-123).  Op level against an f32 restatement on the same
bf16 inputs; module level against the cached-projection path (K13) and against torch's own multi_head_attention_forward arithmetic."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rand(*shape, seed=0, scale=1.0, dt=torch.bfloat16):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV).to(dt)


def _reference(qp, mem, valid):
    """f32: p = softmax_2(qp . mem_j | valid), ctx = sum_j p_j mem_j; exact zeros where no key is valid."""
    B, S, E = mem.shape
    q = qp.float().reshape(B, 8, E)
    s = torch.einsum("bhe,bje->bhj", q, mem.float()) * math.log(2.0)
    s = s.masked_fill(~valid[:, None, :], float("-inf"))
    p = torch.nan_to_num(torch.softmax(s, dim=-1), nan=0.0)
    return torch.einsum("bhj,bje->bhe", p, mem.float()).reshape(B, 8 * E)


@pytest.mark.parametrize("B,S", [(3, 3840), (2, 1000), (5, 96), (300, 160), (1, 8200), (7, 37)])
def test_mqa_decode_matches_f32_reference(B, S):
    """Whole and partial 32-key tiles, one range and several ranges per item (the combine pass), more items than compute units, ragged
    validity incl. an item WITHOUT a valid key and one whose only valid keys sit in the last partial tile."""
    from case_rg_amd import _abi, ops
    E = 512
    qp = _rand(B, 8 * E, seed=1, scale=0.06)   # scores of a few units: a peaked but not one-hot softmax
    mem = _rand(B, S, E, seed=2)
    g = torch.Generator().manual_seed(3)
    valid = (torch.rand(B, S, generator=g) < 0.8).to(DEV)
    valid[0, : S // 3] = True
    if B > 1:
        valid[1] = False            # no valid key: exact zeros
    if B > 2:
        valid[2] = False
        valid[2, S - 1] = True      # a single valid key, the last one: the context IS that row
    got = ops.attention_decode_mqa(qp, mem, valid)
    want = _reference(qp, mem, valid)
    scale = want.abs().max().item()
    err = (got.float() - want).abs().max().item()
    assert err <= 1.2e-2 * scale, "max err %.3e vs scale %.3e" % (err, scale)
    if B > 1:
        assert float(got[1].float().abs().max()) == 0.0
    if B > 2:
        assert torch.equal(got[2].reshape(8, E), mem[2, S - 1].expand(8, E)), "a one-key softmax must return that key's row bit for bit"
    # all keys valid through the null-mask form
    got2 = ops.attention_decode_mqa(qp, mem, None)
    want2 = _reference(qp, mem, torch.ones_like(valid))
    assert (got2.float() - want2).abs().max().item() <= 1.2e-2 * want2.abs().max().item()
    # bit-identical from launch to launch (fixed-order partial sums, no atomics)
    assert torch.equal(got, ops.attention_decode_mqa(qp, mem, valid))
    assert _abi.lib.case_attention_decode_mqa_splits(B, S) >= 1


def test_absorbed_cross_attention_matches_the_cached_projection_path():
    """MultiheadAttention.cross_attention_absorbed (Wk folded into the query, Wv applied to the 512-wide context per head) against
    cross_attention over the cached K / V projections (K13) and against F.multi_head_attention_forward in f32 on the same inputs."""
    import case_rg_amd
    from case_rg_amd.common.attention import MultiheadAttention
    case_rg_amd.set_compute_dtype(torch.bfloat16)
    case_rg_amd.set_dropout(False)
    try:
        torch.manual_seed(5)
        E, N, S = 512, 6, 1500
        m = MultiheadAttention(E, 8).to(DEV).eval()
        with torch.no_grad():
            m.in_proj_bias.copy_(torch.randn(3 * E) * 0.2)
            m.out_proj.bias.copy_(torch.randn(E) * 0.2)
            m.in_proj_weight.mul_(3.0)   # scores of a few units
        case_rg_amd.ops.invalidate_param_cache()
        x = _rand(N, 1, E, seed=11)
        mem = _rand(N, S, E, seed=12)
        valid = torch.ones(N, S, dtype=torch.bool, device=DEV)
        valid[1, 700:] = False
        valid[3, :40] = False
        with torch.no_grad():
            kv = m.project_memory(mem)
            cached = m.cross_attention(x, None, valid, residual=x, kv=kv)
            absorbed = m.cross_attention_absorbed(x, mem, valid, residual=x)
            ref = torch.nn.functional.multi_head_attention_forward(
                x.float().transpose(0, 1), mem.float().transpose(0, 1), mem.float().transpose(0, 1), E, 8,
                m.in_proj_weight.to(torch.bfloat16).float(), m.in_proj_bias, None, None, False, 0.0,
                m.out_proj.weight.to(torch.bfloat16).float(), m.out_proj.bias, training=False, key_padding_mask=~valid,
                need_weights=False)[0].transpose(0, 1) + x.float()
        scale = ref.abs().max().item()
        e_abs, e_cached = (absorbed.float() - ref).abs().max().item(), (cached.float() - ref).abs().max().item()
        assert e_abs <= 2e-2 * scale, "absorbed vs f32 reference: %.3e of %.3e" % (e_abs, scale)
        assert e_abs <= 3.0 * e_cached + 4e-3 * scale, "the absorbed form (%.3e) must be as close to f32 as the cached one (%.3e)" % (e_abs, e_cached)
        # the folded weights follow the parameters: an in-place update (what an optimizer step does) invalidates them
        with torch.no_grad():
            m.in_proj_weight.mul_(0.5)
            again = m.cross_attention_absorbed(x, mem, valid, residual=x)
            kv2 = m.project_memory(mem)
            cached2 = m.cross_attention(x, None, valid, residual=x, kv=kv2)
        assert (again.float() - cached2.float()).abs().max().item() <= 3e-2 * scale
        assert (again.float() - absorbed.float()).abs().max().item() > 1e-3 * scale, "stale absorbed weights"
    finally:
        case_rg_amd.set_compute_dtype(torch.float32)
