"""K20, the fused decoder step chain (csrc/decoder_chain.hip; reference arithmetic: common/TransformerDecoder.py:76-89 on one new position per
sequence, CaSE/Model.py:94-123): the KV-cached greedy step of TransformerDecoder in bf16 at d_model = dim_feedforward = 512 must agree with
(a) the single-launch HIP step on the same weights and caches, position by position, and (b) the f32 CPU oracle's full-prefix decoder
(the reference's O(T^2) form), including batch sizes that are not multiples of the 16-row tile and PAD positions in the history."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture()
def bf16_mode():
    import case_rg_amd
    case_rg_amd.set_compute_dtype(torch.bfloat16)
    case_rg_amd.set_dropout(False)
    yield
    case_rg_amd.set_compute_dtype(torch.float32)


def _decoder(layers, seed):
    import case_rg_amd
    from case_rg_amd.utils import fill_params
    ns = case_rg_amd.namespace()
    layer = ns.TransformerDecoderLayer(512, 8, dim_feedforward=512, dropout=0.1, activation="gelu")
    return fill_params(ns.TransformerDecoder(layer, layers), seed, gain=1.5).to(DEV).eval()


def _run_steps(dec, xs, memory, memory_valid, hist_valid, T):
    """T cached steps over inputs xs [N, T, E]; returns the outputs [N, T, E] and the final self caches."""
    N = xs.shape[0]
    kvs = dec.project_memory(memory)
    cache = dec.new_self_cache(N, T, xs)
    outs = []
    for t in range(T):
        hv = hist_valid.clone()
        hv[:, t + 1:] = False
        outs.append(dec.step(xs[:, t:t + 1].contiguous(), t, cache, hv, kvs, memory_valid))
    return torch.cat(outs, dim=1), cache


@pytest.mark.parametrize("N,S,layers,T", [(37, 200, 2, 6), (256, 64, 4, 3), (5, 77, 1, 9)])
def test_chained_step_matches_the_single_launch_step_and_the_oracle(bf16_mode, N, S, layers, T):
    import oracle
    from case_rg_amd import _abi, ops
    from case_rg_amd.utils import fill_params
    dec = _decoder(layers, 11 + layers)
    g = torch.Generator().manual_seed(N * 100 + S)
    xs = torch.randn(N, T, 512, generator=g)
    mem = torch.randn(N, S, 512, generator=g)
    mlen = torch.randint(S // 2, S + 1, (N,), generator=g)
    mvalid = torch.arange(S)[None, :] < mlen[:, None]
    hist = torch.ones(N, T, dtype=torch.bool)
    hist[N // 2, 1] = False  # a PAD token inside the history of one sequence (tgt_key_padding_mask)
    xb, mb = xs.to(DEV).to(torch.bfloat16), mem.to(DEV).to(torch.bfloat16)
    calls = {}
    raw = _abi.call

    def counting(name, *a):
        calls[name] = calls.get(name, 0) + 1
        return raw(name, *a)

    _abi.call = counting
    ops.DECODER_CHAIN = "on"  # (not the default: slower than the single launches at one row per sequence, see ops.py)
    try:
        with torch.no_grad():
            got, cache = _run_steps(dec, xb, mb, mvalid.to(DEV), hist.to(DEV), T)
            n_chain = calls.get("case_decoder_chain", 0)
            ops.DECODER_CHAIN = "off"
            want_hip, cache_hip = _run_steps(dec, xb, mb, mvalid.to(DEV), hist.to(DEV), T)
    finally:
        ops.DECODER_CHAIN = "off"
        _abi.call = raw
    assert n_chain == T * (2 * layers + 1), "the chain did not run (%d launches)" % n_chain
    assert calls.get("case_decoder_chain", 0) == n_chain, "DECODER_CHAIN = 'off' still launched the chain"
    assert torch.isfinite(got.float()).all()
    # the f32 oracle decodes the whole prefix at once under the causal mask (what the reference's greedy loop re-does every step)
    ref_layer = oracle.TransformerDecoderLayer(512, 8, dim_feedforward=512, dropout=0.1, activation="gelu")
    ref = fill_params(oracle.TransformerDecoder(ref_layer, layers), 11 + layers, gain=1.5).eval()
    causal = torch.triu(torch.full((T, T), float("-inf")), 1)
    with torch.no_grad():
        want = ref(xb.float().cpu().transpose(0, 1), mb.float().cpu().transpose(0, 1), tgt_mask=causal, tgt_key_padding_mask=~hist,
                   memory_key_padding_mask=~mvalid)
        want = (want[0] if isinstance(want, tuple) else want).transpose(0, 1)
    keep = hist.unsqueeze(-1)  # the row of a PAD query is undefined in the reference's own masks as well
    scale = want.abs().max().item()
    err_hip = ((got.float().cpu() - want_hip.float().cpu()) * keep).abs().max().item() / scale
    err_ref = ((got.float().cpu() - want) * keep).abs().max().item() / scale
    base_ref = ((want_hip.float().cpu() - want) * keep).abs().max().item() / scale
    assert err_hip <= 1e-2, "chained vs single-launch step: %.3e" % err_hip
    assert err_ref <= max(2e-2, 1.5 * base_ref), "chained step vs f32 oracle: %.3e (single-launch step: %.3e)" % (err_ref, base_ref)
    for a, b in zip(cache, cache_hip):
        cs = b.float().abs().max().item()
        assert (a.float() - b.float()).abs().max().item() <= 1e-2 * cs, "self-attention K / V cache differs"


def test_chain_entry_point_refuses_what_it_cannot_run(bf16_mode):
    from case_rg_amd import _abi as A
    assert A.lib.case_abi_features() & A.FEAT_DECODER_CHAIN
    x = torch.zeros(16, 512, device=DEV, dtype=torch.bfloat16)
    w = torch.zeros(512, 512, device=DEV, dtype=torch.bfloat16)
    b = torch.zeros(512, device=DEV)
    d = A.DecoderChainDesc()
    d.rows, d.width, d.qkv_parts, d.kv_row_stride = 16, 256, 1, 0
    none = [None] * 10
    with pytest.raises(RuntimeError, match="512"):
        A.call("case_decoder_chain", d, x.data_ptr(), None, w.data_ptr(), b.data_ptr(), *none, x.data_ptr(), None, None, None, 0)
    d.width = 512
    with pytest.raises(RuntimeError, match="nothing to write"):
        A.call("case_decoder_chain", d, x.data_ptr(), None, w.data_ptr(), b.data_ptr(), *([None] * 14), 0)
    with pytest.raises(RuntimeError, match="feed-forward"):  # W1 without W2
        A.call("case_decoder_chain", d, x.data_ptr(), None, w.data_ptr(), b.data_ptr(), b.data_ptr(), b.data_ptr(), w.data_ptr(), b.data_ptr(),
               None, None, None, None, None, None, x.data_ptr(), None, None, None, 0)
