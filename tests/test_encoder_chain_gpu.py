"""K16, the fused encoder chain (csrc/encoder_chain.hip; reference arithmetic: common/TransformerEncoder.py:66-75 per layer): the
inference path of TransformerEncoder / TransformerSeqEncoder in bf16 at d_model 512 must agree with (a) the single-launch HIP
path on the same weights and (b) the f32 CPU oracle, including ragged row counts (tiles of 128 tokens with a partial last tile)
and padded sequences."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _encoder(layers, seed):
    import case_rg_amd
    from case_rg_amd.utils import fill_params
    ns = case_rg_amd.namespace()
    layer = ns.TransformerEncoderLayer(512, 8, dim_feedforward=512, dropout=0.1, activation="gelu")
    return fill_params(ns.TransformerEncoder(layer, layers), seed, gain=2.0).to(DEV).eval()


@pytest.fixture()
def bf16_mode():
    import case_rg_amd
    case_rg_amd.set_compute_dtype(torch.bfloat16)
    case_rg_amd.set_dropout(False)
    yield
    case_rg_amd.set_compute_dtype(torch.float32)


@pytest.mark.parametrize("two_ctx", [False, True])  # round 6: the two-workgroups-per-CU form of the kernel (64-token tiles, four waves) behind CASE_CHAIN_TWO_CTX
@pytest.mark.parametrize("N,L,layers", [(3, 384, 2), (5, 100, 3), (1, 40, 1), (7, 384, 6), (86, 384, 2)])  # the last: 258 tiles = one whole round of full tiles + half tiles
def test_chain_matches_the_single_launch_path_and_the_oracle(bf16_mode, N, L, layers, two_ctx, monkeypatch):
    import oracle
    monkeypatch.setenv("CASE_CHAIN_TWO_CTX", "1" if two_ctx else "0")
    from case_rg_amd import _abi, ops
    from case_rg_amd.utils import fill_params
    enc = _encoder(layers, 31 + layers)
    g = torch.Generator().manual_seed(N * 1000 + L)
    x = torch.randn(N, L, 512, generator=g)
    lens = torch.randint(L // 2, L + 1, (N,), generator=g)
    lens[0] = L
    valid = (torch.arange(L)[None, :] < lens[:, None])
    xb = x.to(DEV).to(torch.bfloat16)
    calls = {}
    raw = _abi.call

    def counting(name, *a):
        calls[name] = calls.get(name, 0) + 1
        return raw(name, *a)

    _abi.call = counting
    try:
        with torch.no_grad():
            got = enc.forward_batch_first(xb, valid.to(DEV))
            n_chain = calls.get("case_encoder_chain", 0)
            ops.ENCODER_CHAIN = "off"
            want_hip = enc.forward_batch_first(xb, valid.to(DEV))
    finally:
        ops.ENCODER_CHAIN = "auto"
        _abi.call = raw
    assert n_chain == layers + 1, "the chain did not run (%d launches)" % n_chain
    assert calls.get("case_encoder_chain", 0) == n_chain, "ENCODER_CHAIN = 'off' still launched the chain"
    ref_layer = oracle.TransformerEncoderLayer(512, 8, dim_feedforward=512, dropout=0.1, activation="gelu")
    ref = fill_params(oracle.TransformerEncoder(ref_layer, layers), 31 + layers, gain=2.0).eval()
    with torch.no_grad():
        want = ref(xb.float().cpu().transpose(0, 1), src_key_padding_mask=~valid).transpose(0, 1)
    m = valid.unsqueeze(-1)
    scale = want.abs().max().item()
    err_hip = ((got.float().cpu() - want_hip.float().cpu()) * m).abs().max().item() / scale
    err_ref = ((got.float().cpu() - want) * m).abs().max().item() / scale
    base_ref = ((want_hip.float().cpu() - want) * m).abs().max().item() / scale
    assert torch.isfinite(got.float()).all()
    assert err_ref <= max(2e-2, 1.5 * base_ref), "chain vs f32 oracle: %.3e (single-launch path: %.3e)" % (err_ref, base_ref)
    assert err_hip <= 2e-2, "chain vs single-launch HIP path: %.3e" % err_hip


def test_chain_pack_follows_parameter_updates(bf16_mode):
    from case_rg_amd import ops
    enc = _encoder(2, 77)
    x = torch.randn(2, 128, 512, device=DEV).to(torch.bfloat16)
    with torch.no_grad():
        a = enc.forward_batch_first(x)
        enc.layers[1].linear1.weight.mul_(0.5)  # in place: _version moves, the cached pack must not be served
        b = enc.forward_batch_first(x)
        ops.ENCODER_CHAIN = "off"
        try:
            c = enc.forward_batch_first(x)
        finally:
            ops.ENCODER_CHAIN = "auto"
    assert not torch.equal(a, b)
    assert (b.float() - c.float()).abs().max().item() <= 2e-2 * c.float().abs().max().item()


def test_seq_encoder_eval_uses_the_chain_and_training_does_not(bf16_mode):
    import case_rg_amd
    from case_rg_amd import _abi
    from case_rg_amd.utils import fill_params
    ns = case_rg_amd.namespace()
    enc = fill_params(ns.TransformerSeqEncoder(3, 8, 1000, 512), 5).to(DEV)
    ids = torch.randint(1, 1000, (2, 3, 64), device=DEV)
    calls = {}
    raw = _abi.call

    def counting(name, *a):
        calls[name] = calls.get(name, 0) + 1
        return raw(name, *a)

    _abi.call = counting
    try:
        enc.eval()
        with torch.no_grad():
            y_eval = enc(ids)[0]
        n_eval = calls.get("case_encoder_chain", 0)
        enc.train()
        y_train = enc(ids)[0]
        y_train.float().sum().backward()
    finally:
        _abi.call = raw
    assert n_eval == 4 and calls.get("case_encoder_chain", 0) == 4, calls.get("case_encoder_chain", 0)
    assert (y_eval.float() - y_train.float()).abs().max().item() <= 3e-2 * y_train.float().abs().max().item()


def test_shared_encoder_runs_query_and_passages_in_one_pass(bf16_mode):
    """TransformerSeqEncoder.forward_many (training, bf16): the query rows ride along with the passage rows through every row-local
    launch; outputs equal the two separate passes (a GEMM row does not depend on its neighbours), parameter gradients agree up to the
    summation order of the weight gradients, and the launch count of the row-local work is that of ONE pass."""
    import case_rg_amd
    from case_rg_amd import _abi
    from case_rg_amd.utils import fill_params
    ns = case_rg_amd.namespace()
    enc = fill_params(ns.TransformerSeqEncoder(2, 8, 500, 512), 5, gain=2.0).to(DEV).train()
    g = torch.Generator().manual_seed(3)
    B, P, Lp, Lq = 4, 3, 128, 64
    query = torch.randint(1, 500, (B, 1, Lq), generator=g)
    passage = torch.randint(1, 500, (B, P, Lp), generator=g)
    query[:, :, 50:] = 0
    passage[1, 2, 2:] = 0
    query, passage = query.to(DEV), passage.to(DEV)
    wq = torch.randn(B, 1, 1, Lq, 512, generator=g).to(DEV)
    wp = torch.randn(B, P, 1, Lp, 512, generator=g).to(DEV)
    calls, raw = {}, _abi.call

    def counting(name, *a):
        calls[name] = calls.get(name, 0) + 1
        return raw(name, *a)

    res = []
    for merged in (True, False):
        enc.zero_grad()
        calls.clear()
        _abi.call = counting
        try:
            if merged:
                (oq, sq), (op, sp) = enc.forward_many([query, passage])
            else:
                (oq, sq), (op, sp) = enc(query), enc(passage)
            loss = (oq.float() * wq).sum() + (op.float() * wp).sum() + sq.float().sum() + sp.float().sum()
            loss.backward()
        finally:
            _abi.call = raw
        res.append((oq.detach().float(), op.detach().float(), sq.detach().float(), sp.detach().float(),
                    {n: p.grad.detach().clone() for n, p in enc.named_parameters() if p.grad is not None}, dict(calls)))
    (oq1, op1, sq1, sp1, g1, c1), (oq0, op0, sq0, sp0, g0, c0) = res
    assert torch.equal(oq1, oq0) and torch.equal(op1, op0) and torch.equal(sq1, sq0) and torch.equal(sp1, sp0)
    assert set(g1) == set(g0)
    for n in g0:
        denom = g0[n].norm().clamp_min(1e-12)
        assert ((g1[n] - g0[n]).norm() / denom).item() < 2e-3, n
    assert c1["case_attention_fwd"] == c0["case_attention_fwd"] == 4 and c1["case_attention_bwd"] == 4   # two groups x two layers
    assert c1["case_layernorm_fwd"] * 2 == c0["case_layernorm_fwd"] and c1["case_gemm"] < 0.62 * c0["case_gemm"]
