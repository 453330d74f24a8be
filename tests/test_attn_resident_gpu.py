"""K18 (csrc/attn64.hip): head_dim-64 attention with the (sequence, head) resident in one twelve-wave workgroup -- the forward
and (later) the single-pass backward against an f32 restatement of F.multi_head_attention_forward's bmm -> softmax -> bmm chain on
the same bf16 inputs (common/TransformerEncoder.py:67, common/TransformerBlock.py:26), and against the flash-style kernels of
attention.hip it replaces (identical dropout masks)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _ops():
    from case_rg_amd import ops
    return ops


def _reference(qkv, valid, h, d):
    N, L, _ = qkv.shape
    E = h * d
    q, k, v = qkv.float().split(E, dim=-1)
    qh, kh, vh = [t.reshape(N, L, h, d).transpose(1, 2) for t in (q, k, v)]
    s = (qh @ kh.transpose(-1, -2)) / math.sqrt(d)
    s = s.masked_fill(~valid[:, None, None, :], float("-inf"))
    p = torch.softmax(s, -1).nan_to_num(0.0)
    return (p @ vh).transpose(1, 2).reshape(N, L, E), torch.logsumexp(s, -1)


def _rel(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-20)).item()


@pytest.mark.parametrize("N,h,L", [(2, 8, 384), (3, 8, 320), (1, 2, 292), (40, 8, 384), (33, 3, 352)])
def test_resident_forward_matches_reference(N, h, L):
    """Full and ragged validity (one sequence cut in the second chunk, one with a single valid key, one with none), fewer and more
    items than workgroups (40 x 8 = 320 items on 256 persistent workgroups: the DMA stream crosses item boundaries)."""
    from case_rg_amd import _abi
    ops = _ops()
    d, E = 64, h * 64
    ad = _abi.AttnDesc()
    g = torch.Generator().manual_seed(L + N)
    qkv = (torch.randn(N, L, 3 * E, generator=g) * 0.7).to(DEV).to(torch.bfloat16)
    valid = torch.ones(N, L, dtype=torch.bool, device=DEV)
    valid[N - 1, L // 2 + 3:] = False
    if N > 2:
        valid[1, 1:] = False
        valid[2, :] = False
        valid[N - 2] = (torch.rand(L, generator=g) > 0.3).to(DEV)
    o = ops.attention(qkv, qkv, qkv, 0, E, 2 * E, h, d, key_valid=valid)
    ref, _ = _reference(qkv, valid, h, d)
    err = _rel(o, ref)
    assert err < 6e-3, err
    if N > 2:
        assert torch.count_nonzero(o[2]) == 0, "no valid key: exact zeros"
    assert torch.isfinite(o.float()).all()


def test_resident_forward_equals_flash_kernels_with_dropout():
    """Same counter RNG and element index as the flash-style kernel and the unfused softmax: the dropped-out outputs agree to bf16
    rounding of P (the masks are identical); LSE agrees to f32 rounding."""
    from case_rg_amd import _abi, config
    ops = _ops()
    N, h, L, d = 5, 8, 384, 64
    E = h * d
    g = torch.Generator().manual_seed(5)
    qkv = (torch.randn(N, L, 3 * E, generator=g) * 0.7).to(DEV).to(torch.bfloat16)
    valid = torch.ones(N, L, dtype=torch.bool, device=DEV)
    valid[3, 200:] = False
    config.set_dropout(True)
    try:
        outs = []
        for unfused in (False, True):
            config.manual_seed(11)
            saved = _abi.lib.case_attention_supported
            try:
                if unfused:
                    _abi.lib.case_attention_supported = lambda _d: 0
                outs.append(ops.attention(qkv, qkv, qkv, 0, E, 2 * E, h, d, key_valid=valid, p_drop=0.1))
            finally:
                _abi.lib.case_attention_supported = saved
        err = _rel(outs[0], outs[1])
        assert err < 8e-3, err
    finally:
        config.set_dropout(False)


def _grads_reference(qkv, valid, h, d, g):
    N, L, _ = qkv.shape
    E = h * d
    r = qkv.detach().float().requires_grad_()
    q, k, v = r.split(E, dim=-1)
    qh, kh, vh = [t.reshape(N, L, h, d).transpose(1, 2) for t in (q, k, v)]
    s = (qh @ kh.transpose(-1, -2)) / math.sqrt(d)
    s = s.masked_fill(~valid[:, None, None, :], float("-inf"))
    p = torch.softmax(s, -1).nan_to_num(0.0)
    (p @ vh).transpose(1, 2).reshape(N, L, E).backward(g.float())
    return r.grad


@pytest.mark.parametrize("N,h,L", [(2, 8, 384), (3, 8, 320), (1, 2, 292), (40, 8, 384), (33, 3, 352)])
def test_resident_backward_matches_reference(N, h, L):
    """K19, the single-pass backward (dK / dV stationary, dQ through the dS^T tile in LDS): dQ, dK, dV against the f32 autograd of the
    same bf16 inputs; ragged validity, a sequence without a valid key (zero gradients), workgroups with one and with two items."""
    ops = _ops()
    d, E = 64, h * 64
    g0 = torch.Generator().manual_seed(7 * L + N)
    qkv = (torch.randn(N, L, 3 * E, generator=g0) * 0.7).to(DEV).to(torch.bfloat16).requires_grad_()
    g = (torch.randn(N, L, E, generator=g0)).to(DEV).to(torch.bfloat16)
    valid = torch.ones(N, L, dtype=torch.bool, device=DEV)
    valid[N - 1, L // 2 + 3:] = False
    if N > 2:
        valid[1, 1:] = False
        valid[2, :] = False
        valid[N - 2] = (torch.rand(L, generator=g0) > 0.3).to(DEV)
    ops.attention(qkv, qkv, qkv, 0, E, 2 * E, h, d, key_valid=valid).backward(g)
    ref = _grads_reference(qkv, valid, h, d, g)
    got = qkv.grad.float()
    assert torch.isfinite(got).all()
    for name, sl in (("dq", slice(0, E)), ("dk", slice(E, 2 * E)), ("dv", slice(2 * E, 3 * E))):
        err = _rel(got[..., sl], ref[..., sl])
        assert err < 1.2e-2, (name, err)
    if N > 2:
        assert torch.count_nonzero(got[2]) == 0, "no valid key: zero gradients"


def test_resident_backward_equals_unfused_path_with_dropout():
    """Dropout: the backward regenerates the forward's mask (same counter RNG / element index as the unfused GEMM + softmax path)."""
    from case_rg_amd import _abi, config
    ops = _ops()
    N, h, L, d = 5, 8, 384, 64
    E = h * d
    g0 = torch.Generator().manual_seed(5)
    base = (torch.randn(N, L, 3 * E, generator=g0) * 0.7).to(DEV).to(torch.bfloat16)
    g = (torch.randn(N, L, E, generator=g0)).to(DEV).to(torch.bfloat16)
    valid = torch.ones(N, L, dtype=torch.bool, device=DEV)
    valid[3, 200:] = False
    config.set_dropout(True)
    try:
        grads = []
        for unfused in (False, True):
            config.manual_seed(11)
            x = base.clone().requires_grad_()
            saved = _abi.lib.case_attention_supported
            try:
                if unfused:
                    _abi.lib.case_attention_supported = lambda _d: 0
                ops.attention(x, x, x, 0, E, 2 * E, h, d, key_valid=valid, p_drop=0.1).backward(g)
            finally:
                _abi.lib.case_attention_supported = saved
            grads.append(x.grad.float())
        for name, sl in (("dq", slice(0, E)), ("dk", slice(E, 2 * E)), ("dv", slice(2 * E, 3 * E))):
            err = _rel(grads[0][..., sl], grads[1][..., sl])
            assert err < 2e-2, (name, err)
    finally:
        config.set_dropout(False)


def _cross_reference(q, kv, valid, h, d, g=None):
    """f32 attention of q [N, Lq, E] over kv [N, Lk, 2E] (K | V); with ``g`` also the gradients of q and kv."""
    N, Lq, E = q.shape
    Lk = kv.shape[1]
    qf, kvf = q.detach().float().requires_grad_(g is not None), kv.detach().float().requires_grad_(g is not None)
    k, v = kvf.split(E, dim=-1)
    qh = qf.reshape(N, Lq, h, d).transpose(1, 2)
    kh, vh = [t.reshape(N, Lk, h, d).transpose(1, 2) for t in (k, v)]
    s = (qh @ kh.transpose(-1, -2)) / math.sqrt(d)
    s = s.masked_fill(~valid[:, None, None, :], float("-inf"))
    p = torch.softmax(s, -1).nan_to_num(0.0)
    o = (p @ vh).transpose(1, 2).reshape(N, Lq, E)
    if g is None:
        return o
    o.backward(g.float())
    return o, qf.grad, kvf.grad


@pytest.mark.parametrize("N,h,Lq,Lk", [(3, 8, 40, 320), (2, 4, 300, 384), (5, 8, 257, 292), (2, 8, 384, 384), (4, 8, 1, 384), (2, 8, 64, 289 + 3)])
def test_resident_kernels_with_separate_query_and_memory_tensors(N, h, Lq, Lk):
    """The same kernels behind the cross-attention call form: q from its own [N, Lq, E] tensor, K | V packed in a [N, Lk, 2E] memory
    projection (different row strides for q, k / v and the output), Lq != Lk -- query counts that fill only part of a wave group, the
    shortest and the longest key counts the resident kernels take, masked keys.  Forward for every shape; the backward kernel takes
    Lq > 256 (the others run the flash-style backward on the same forward output)."""
    ops = _ops()
    d, E = 64, h * 64
    g0 = torch.Generator().manual_seed(Lq * 1000 + Lk)
    q = (torch.randn(N, Lq, E, generator=g0) * 0.7).to(DEV).to(torch.bfloat16).requires_grad_()
    kv = (torch.randn(N, Lk, 2 * E, generator=g0) * 0.7).to(DEV).to(torch.bfloat16).requires_grad_()
    g = torch.randn(N, Lq, E, generator=g0).to(DEV).to(torch.bfloat16)
    valid = torch.ones(N, Lk, dtype=torch.bool, device=DEV)
    valid[0, Lk - 5:] = False
    valid[N - 1] = (torch.rand(Lk, generator=g0) > 0.4).to(DEV)
    o = ops.attention(q, kv, kv, 0, 0, E, h, d, key_valid=valid)
    o.backward(g)
    ref_o, ref_dq, ref_dkv = _cross_reference(q, kv, valid, h, d, g)
    assert torch.isfinite(o.float()).all() and torch.isfinite(q.grad.float()).all() and torch.isfinite(kv.grad.float()).all()
    assert _rel(o, ref_o) < 6e-3
    assert _rel(q.grad, ref_dq) < 1.2e-2
    assert _rel(kv.grad[..., :E], ref_dkv[..., :E]) < 1.2e-2 and _rel(kv.grad[..., E:], ref_dkv[..., E:]) < 1.2e-2


def test_resident_kernels_without_a_key_mask():
    """key_valid = None (every key counts): forward and backward against the f32 reference."""
    ops = _ops()
    N, h, L, d = 3, 8, 352, 64
    E = h * d
    g0 = torch.Generator().manual_seed(99)
    qkv = (torch.randn(N, L, 3 * E, generator=g0) * 0.7).to(DEV).to(torch.bfloat16).requires_grad_()
    g = torch.randn(N, L, E, generator=g0).to(DEV).to(torch.bfloat16)
    ops.attention(qkv, qkv, qkv, 0, E, 2 * E, h, d, key_valid=None).backward(g)
    valid = torch.ones(N, L, dtype=torch.bool, device=DEV)
    with torch.no_grad():
        o = ops.attention(qkv, qkv, qkv, 0, E, 2 * E, h, d, key_valid=None)
    ref_o, _ = _reference(qkv.detach(), valid, h, d)
    ref = _grads_reference(qkv, valid, h, d, g)
    assert _rel(o, ref_o) < 6e-3
    for name, sl in (("dq", slice(0, E)), ("dk", slice(E, 2 * E)), ("dv", slice(2 * E, 3 * E))):
        assert _rel(qkv.grad.float()[..., sl], ref[..., sl]) < 1.2e-2, name


def test_resident_kernels_are_bit_reproducible_across_launches():
    """K18 / K19 have no atomics and a static DMA schedule: the same inputs must give the same bits, launch after launch, with other
    work in between (a race between a DMA and its consumer would show as a rare difference) -- 24 rounds on fresh inputs, dropout on."""
    from case_rg_amd import config
    ops = _ops()
    N, h, L, d = 36, 8, 384, 64  # 288 items on 256 persistent workgroups: item boundaries inside a workgroup
    E = h * d
    config.set_dropout(True)
    try:
        for rnd in range(24):
            g0 = torch.Generator().manual_seed(1000 + rnd)
            base = (torch.randn(N, L, 3 * E, generator=g0) * 0.7).to(DEV).to(torch.bfloat16)
            g = torch.randn(N, L, E, generator=g0).to(DEV).to(torch.bfloat16)
            valid = torch.ones(N, L, dtype=torch.bool, device=DEV)
            valid[rnd % N, 100 + 4 * rnd:] = False
            outs = []
            for rep in range(2):
                config.manual_seed(77 + rnd)
                x = base.clone().requires_grad_()
                o = ops.attention(x, x, x, 0, E, 2 * E, h, d, key_valid=valid, p_drop=0.1)
                if rep == 0:  # unrelated work between the two launches
                    torch.mm(base[0].float(), base[1].float().t())
                o.backward(g)
                outs.append((o.detach().clone(), x.grad.clone()))
            assert torch.equal(outs[0][0], outs[1][0]), "forward differs between two launches (round %d)" % rnd
            assert torch.equal(outs[0][1], outs[1][1]), "backward differs between two launches (round %d)" % rnd
    finally:
        config.set_dropout(False)
