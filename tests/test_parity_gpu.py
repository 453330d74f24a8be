"""Parity proper (-m gpu): the HIP product, called through the reference-named modules (which reach the
kernels through the C ABI), against (a) the committed golden fixtures captured from the reference itself and
(b) the CPU oracle run on the same seeded inputs.

fp32 compute mode, dropout off (torch's dropout RNG stream cannot be reproduced; the reference fixtures were
captured with dropout patched to identity).  Bar: 1e-3 relative to each tensor's scale (north star); greedy token
ids exact wherever the reference's own top1-top2 margin exceeds 1e-3."""
import numpy as np
import pytest
import torch

import cases
from helpers import load_golden, record_error, scaled_error, to_np

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ns():
    import case_rg_amd
    case_rg_amd.set_compute_dtype(torch.float32)
    case_rg_amd.set_dropout(False)
    return case_rg_amd.namespace()


def _scaled_close(name, got, want, tol):
    case, _, key = name.partition("/")
    rel = scaled_error(name, got, want)
    record_error(case, "fp32", key, rel, tol)
    assert rel <= tol, "%s: %.2e relative to the tensor's scale (tol %.0e)" % (name, rel, tol)


# bars: 1e-3 of each tensor's scale (north star), gradients included.  Weights / EMA after Adam steps get 5e-3: Adam divides
# by sqrt(v), which turns a 1e-4 relative gradient difference on a near-zero gradient element into a visible step difference.
def _check(name, rec, tol=1e-3, grad_tol=1e-3):
    golden = load_golden(name)
    assert set(rec) == set(golden), "case %s: keys differ: %s" % (name, set(rec) ^ set(golden))
    for k, want in golden.items():
        got = to_np(rec[k])
        if want.dtype.kind in "biu":
            if k in ("answer",):
                continue  # checked with the margin rule below
            assert np.array_equal(got, want), "%s/%s: integer mismatch" % (name, k)
        elif k == "margin":
            continue
        elif k.startswith(("w_slice", "ema_slice")):
            _scaled_close(name + "/" + k, got, want, 5e-3)
        elif name == "trainer_traj" and k == "rank":
            _scaled_close(name + "/" + k, got, want, 2e-2)  # scores of the model AFTER the Adam steps (see above)
        else:
            _scaled_close(name + "/" + k, got, want, grad_tol if k.startswith("g") else tol)


MODULE_CASES = [n for n in cases.CASES if n not in cases.MODEL_CASES + cases.PROD_CASES + cases.PROD_FORWARD_CASES + cases.PROD_TEST_CASES]


@pytest.mark.parametrize("name", MODULE_CASES)
def test_module_matches_reference_fixture(ns, name):
    rec = cases.CASES[name](ns, torch.device("cuda"))
    _check(name, rec)


@pytest.mark.parametrize("name", ["case_train", "masque_train"])
def test_training_losses_and_gradients_match_reference(ns, name):
    rec = cases.CASES[name](ns, torch.device("cuda"))
    _check(name, rec)


@pytest.mark.parametrize("name", ["case_test", "masque_test"])
def test_greedy_ids_exact_and_rank_scores(ns, name):
    rec = cases.CASES[name](ns, torch.device("cuda"))
    golden = load_golden(name)
    _scaled_close(name + "/rank", to_np(rec["rank"]), golden["rank"], 1e-3)
    got, want, margin = to_np(rec["answer"]), golden["answer"], golden["margin"]
    assert got.shape == want.shape
    checked = 0
    for b in range(want.shape[0]):
        for t in range(want.shape[1]):
            if margin[b, t] <= 1e-3:
                break  # a near-tie may legitimately flip; later steps then see another prefix
            assert got[b, t] == want[b, t], "%s: token (%d,%d) %d != reference %d (margin %.3g)" % (
                name, b, t, got[b, t], want[b, t], margin[b, t])
            checked += 1
    assert checked >= want.size // 2, "too few decisive positions were checked"
    _scaled_close(name + "/margin", to_np(rec["margin"]), margin, 2e-3)


def test_product_matches_oracle_on_fresh_inputs(ns):
    """Not a fixture replay: a new seed, the oracle computed live on the host, same filler."""
    import oracle
    from case_rg_amd.utils import fill_params, make_vocab, synth_batch
    v2i, i2v = make_vocab(300)
    b = synth_batch(3, 4, 20, 10, 8, 300, seed=999, model="case")
    ref = fill_params(oracle.CaSE(4, 8, i2v, v2i, 64), 5).train()
    prod = fill_params(ns.CaSE(4, 8, i2v, v2i, 64), 5).cuda().train()
    want = ref(dict(b), method="train")
    got = prod({k: v.cuda() for k, v in b.items()}, method="train")
    for w, g in zip(want, got):
        assert abs(w.item() - g.item()) <= 1e-3 * max(1.0, abs(w.item()))
    sum(w.mean() for w in want).backward()
    sum(g.mean() for g in got).backward()
    rp, pp = dict(ref.named_parameters()), dict(prod.named_parameters())
    worst, worst_name = 0.0, ""
    for n_, p in rp.items():
        g = pp[n_].grad
        assert g is not None, "no gradient for " + n_
        scale = p.grad.abs().max().item() + 1e-8
        rel = (g.cpu() - p.grad).abs().max().item() / scale
        if rel > worst:
            worst, worst_name = rel, n_
    record_error("fresh_inputs_vs_oracle", "fp32", "worst_gradient:" + worst_name, worst, 2e-3)
    assert worst <= 2e-3, "worst relative gradient error %.3e (%s)" % (worst, worst_name)


def test_bf16_mode_runs_and_is_close(ns):
    """Throughput mode (bf16 storage / bf16 MFMA, f32 accumulate): reported against its own, looser bar."""
    import case_rg_amd
    case_rg_amd.set_compute_dtype(torch.bfloat16)
    try:
        rec = cases.CASES["case_train"](ns, torch.device("cuda"))
        golden = load_golden("case_train")
        for k in ("loss_ps", "loss_se", "loss_rg"):
            assert abs(float(to_np(rec[k])[0]) - float(golden[k][0])) <= 3e-2 * max(1.0, abs(float(golden[k][0]))), k
    finally:
        case_rg_amd.set_compute_dtype(torch.float32)


def test_reference_checkpoint_schema_loads_strict(ns):
    import oracle
    from case_rg_amd.utils import fill_params, make_vocab
    v2i, i2v = make_vocab(200)
    src = fill_params(oracle.CaSE(4, 6, i2v, v2i, 32), 3)
    dst = ns.CaSE(4, 6, i2v, v2i, 32)
    dst.load_state_dict(src.state_dict(), strict=True)


def test_error_behaviour_mirrors_the_reference(ns):
    """RuntimeError for an unknown activation (TransformerEncoder.py:17), assertion for num_q not in {1, num_p}
    (Interaction.py:27), and a loud error instead of an out-of-bounds read past the sinusoid table (max_len 1000)."""
    with pytest.raises(RuntimeError, match="relu/gelu"):
        ns.TransformerEncoderLayer(32, 8, 32, activation="swish")
    inter = ns.Interaction(32).cuda()
    e2 = torch.zeros(2, 2, 4, 32, device="cuda")
    e3 = torch.zeros(2, 3, 5, 32, device="cuda")
    with pytest.raises(AssertionError):
        inter(e2, e3, torch.ones(2, 2, 4, dtype=torch.bool, device="cuda"), torch.ones(2, 3, 5, dtype=torch.bool, device="cuda"))
    enc = ns.TransformerSeqEncoder(1, 8, 120, 32).cuda()
    with pytest.raises(RuntimeError, match="max_len"):
        enc(torch.ones(1, 1, 1001, dtype=torch.long, device="cuda"))


def test_full_size_properties_cfg2_shapes(ns):
    """Size-independent properties at the BASELINE cfg 2 shapes (one batch item, 10 x 384 passages, H = 512, bf16):
    padded positions come out exactly zero, the copy distribution carries probability mass only on source ids, every
    distribution row sums to 1, and appending padding to every passage does not change the losses."""
    import case_rg_amd
    from case_rg_amd.utils import fill_params, make_vocab, synth_batch
    case_rg_amd.set_compute_dtype(torch.bfloat16)
    try:
        V = 2000
        v2i, i2v = make_vocab(V)
        model = fill_params(ns.CaSE(4, 40, i2v, v2i, 512), 3).cuda().train()
        b = {k: v.cuda() for k, v in synth_batch(1, 10, 384, 64, 40, V, seed=5, model="case").items()}
        with torch.no_grad():
            eq, ep = model.query_encoder(b["query"]), model.passage_encoder(b["passage"])
            ps = model.passage_selection.action(b["query"], b["passage"], encode_query=eq, encode_passage=ep)
            se = model.span_extraction.action(b["query"], b["passage"], encode_query=eq, encode_passage=ep, passage_selection_result=ps)
            pad = b["passage"].eq(0)
            assert pad.any() and (ps[2][0][pad] == 0).all(), "selection-stage reps must be exactly zero at pads"
            rg = model.response_generation.action(b["query"], b["passage"], b["source_map"], eq, ep, ps, se, output=b["response"])
            d1, d2 = rg[2]
            total = (d1 + d2).sum(-1)
            assert torch.allclose(total, torch.ones_like(total), atol=2e-2), "distribution rows must sum to 1"
            off_source = torch.ones(V, dtype=torch.bool, device="cuda")
            off_source[b["source_map"][0]] = False
            assert (d2[0][:, off_source] == 0).all(), "copy mass outside the source ids"
            base = [l.item() for l in model(dict(b), method="train")]
            wider = dict(b)
            wider["passage"] = torch.nn.functional.pad(b["passage"], (0, 16))
            wider["token_label"] = torch.nn.functional.pad(b["token_label"], (0, 16))
            wider["token_weight"] = torch.nn.functional.pad(b["token_weight"], (0, 16), value=1.0)
            wider["source_map"] = torch.cat([b["query"].reshape(1, -1), wider["passage"].reshape(1, -1)], 1)
            more = [l.item() for l in model(wider, method="train")]
            for x, y in zip(base, more):
                assert abs(x - y) <= 3e-2 * max(1.0, abs(x)), (base, more)
    finally:
        case_rg_amd.set_compute_dtype(torch.float32)


@pytest.mark.parametrize("second_stream", [False, True])
def test_gradsync_over_rccl_on_one_rank(ns, second_stream):
    """Drives the data-parallel machinery (post-accumulate hooks -> flat buckets -> asynchronous RCCL all-reduce ->
    scatter back) on a one-rank "nccl" group on the GPU: the synchronised gradients must equal the plain ones.
    ``second_stream``: with the query-side block stacks on their own stream (common/heads.run_block_pair; forced here, the policy reserves it for
    GPU-bound geometries) the hooks fire on two streams -- GradSync joins them before it gathers a bucket (ops.join_aux_streams)."""
    import socket
    from case_rg_amd.common import heads
    keep_min = heads.SIDE_STREAM_MIN_ELEMS
    keep_dp = heads.SIDE_STREAM_NOT_UNDER_DP
    if second_stream:
        heads.SIDE_STREAM_MIN_ELEMS = 0
        heads.SIDE_STREAM_NOT_UNDER_DP = False  # (the policy keeps the second stream out of data-parallel runs: the reserved CUs; forced here)
    import torch.distributed as dist
    from case_rg_amd.parallel import GradSync
    from case_rg_amd.utils import fill_params, make_vocab, synth_batch
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        v2i, i2v = make_vocab(300)
        model = fill_params(ns.Masque(6, i2v, v2i, 64), 4).cuda().train()
        b = {k: v.cuda() for k, v in synth_batch(2, 3, 16, 8, 6, 300, seed=8, model="masque").items()}
        sum(l.mean() for l in model(dict(b), method="train")).backward()
        plain = {n: p.grad.clone() for n, p in model.named_parameters()}
        model.zero_grad()
        sync = GradSync(model, bucket_mb=0.25, force=True)
        assert sync.active and len(sync.buckets) > 2
        sum(l.mean() for l in model(dict(b), method="train")).backward()
        sync.finish()
        torch.cuda.synchronize()
        # f32 atomics (embedding rows, split-K weight gradients) make two runs differ in the last bits
        for n, p in model.named_parameters():
            scale = plain[n].abs().max().item() + 1e-12
            err = (p.grad - plain[n]).abs().max().item()
            assert err <= 1e-4 * scale, "%s: %.3e vs scale %.3e" % (n, err, scale)
        assert bool(heads._side) or not (second_stream and heads.SIDE_STREAM), "the second stream was never used"
    finally:
        heads.SIDE_STREAM_MIN_ELEMS, heads.SIDE_STREAM_NOT_UNDER_DP = keep_min, keep_dp
        dist.destroy_process_group()


def test_sentences_on_device_match_reference_fixture(ns):
    """to_sentence through the device compaction kernel (ids on the GPU) + remove_duplicate vs the reference's own output."""
    rec = cases.CASES["sentences"](ns, torch.device("cuda"))
    golden = load_golden("sentences")
    for k in ("sentences", "deduplicated"):
        assert np.array_equal(to_np(rec[k]), golden[k]), k


def test_greedy_early_stop_and_graph_replay(ns):
    """(1) EOS-aware early stop: same sentences as the reference's fixed-T loop, fewer steps, PAD behind finished answers;
    (2) one whole greedy pass captured in a hipGraph replays to identical ids (cfg 4's captured-step requirement)."""
    from case_rg_amd.utils import fill_params, make_vocab, synth_batch
    V_, T = 200, 24
    v2i, i2v = make_vocab(V_)
    model = fill_params(ns.CaSE(4, T, i2v, v2i, 32), 153, gain=3.0).cuda().eval()
    b = {k: v.cuda() for k, v in synth_batch(4, 3, 12, 8, 6, V_, seed=152, model="case").items()}
    dec = model.response_generation.decoder
    with torch.no_grad():
        full = model(dict(b), method="test")["answer"]
        assert dec.last_greedy_steps == T
        # graph capture of the same pass (fixed T steps inside a capture): replay must reproduce the eager ids
        static_out = {}
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            model(dict(b), method="test")  # warm-up on the capture stream
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            static_out["answer"] = model(dict(b), method="test")["answer"]
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(static_out["answer"], full), "hipGraph replay of the greedy pass differs from the eager pass"
        # declare the most frequent generated token to be EOS and keep the batch rows that emit it (items are independent),
        # so that every remaining answer finishes before T
        eos = int(torch.mode(full.reshape(-1)).values)
        rows = [r for r in range(full.size(0)) if (full[r] == eos).any()]
        assert rows, "no row emits the chosen token"
        sub = {k: v[rows] for k, v in b.items()}
        full = model(dict(sub), method="test")["answer"]
        ends = [int((row == eos).nonzero()[0]) for row in full]
        dec.eos_id, dec.eos_check_every = eos, 4
        try:
            early = model(dict(sub), method="test")["answer"]
        finally:
            dec.eos_id = None
        assert dec.last_greedy_steps <= min(T, ((max(ends) + 1 + 3) // 4) * 4), "the loop did not stop after the last EOS"
        if max(ends) + 1 <= T - 4:
            assert dec.last_greedy_steps < T
        for r, cut in enumerate(ends):
            assert torch.equal(early[r, :cut + 1], full[r, :cut + 1]) and (early[r, cut + 1:] == 0).all()
        i2v_eos = dict(i2v)
        i2v_eos[v2i["[unused1]"]], i2v_eos[eos] = "tok_old_eos", "[unused1]"
        assert ns.to_sentence(early, i2v_eos) == ns.to_sentence(full, i2v_eos)


def _sharpen_heads(model, v_gain=40.0, gen_gain=4.0):
    """Decisive decoding without training: scale the last linear maps in front of the softmaxes (pointer heads' ``v``, vocabulary
    projection), as the production-geometry greedy fixtures do (tests/golden/cases.py: PROD_TEST_GAIN).  The top-1 / top-2 margins
    then sit well above bf16 resolution, which is the property a trained model has and a plain random one lacks."""
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.startswith("response_generation.decoder.attns.") and n.endswith(".v.weight"):
                p.mul_(v_gain)
            if n in ("response_generation.decoder.gen.2.weight", "response_generation.decoder.gen.1.weight"):
                p.mul_(gen_gain)
    return model


def test_rouge_l_of_greedy_answers_within_0p2_of_the_oracle(ns):
    """North-star acceptance in miniature ("ROUGE-L on dev within 0.2 of reference"): a synthetic dev set is decoded greedily by the
    HIP model (fp32 AND bf16) and by the CPU oracle with the same weights; ROUGE-L (the reference's metric, pinned by
    tests/golden/rouge_l.npz) of each against the ground-truth answers must agree within 0.2 points in BOTH modes.  The weights are
    random with sharpened output heads (see _sharpen_heads): a plain random model has top-1 / top-2 margins below bf16 resolution
    (round 2: 0.31 points in bf16); training the oracle on a synthetic copy task for 200-300 Adam steps in the build container did
    not converge to a decoding that differs from "uniform over the query words" within the time a test may take (recorded in
    DESIGN.md), so the decisiveness is set directly instead of learned."""
    import case_rg_amd
    import oracle
    from case_rg_amd.evaluation import eval_rouge_l
    from case_rg_amd.utils import fill_params, make_vocab, synth_batch
    V_, T = 300, 12
    v2i, i2v = make_vocab(V_)
    batch = synth_batch(96, 3, 16, 8, T, V_, seed=777, model="case")
    truth = [" ".join(w) for w in oracle.to_sentence(batch["response"].tolist(), i2v)]
    ref_model = _sharpen_heads(fill_params(oracle.CaSE(4, T, i2v, v2i, 64), 21, gain=3.0)).eval()
    with torch.no_grad():
        ans = ref_model(dict(batch), method="test")["answer"]
    want = eval_rouge_l([" ".join(w) for w in oracle.to_sentence(ans.tolist(), i2v)], [[t] for t in truth])
    scores, same, prefix = {}, {}, {}
    for dt in (torch.float32, torch.bfloat16):
        case_rg_amd.set_compute_dtype(dt)
        try:
            model = _sharpen_heads(fill_params(ns.CaSE(4, T, i2v, v2i, 64), 21, gain=3.0)).cuda().eval()
            with torch.no_grad():
                got = model({k: v.cuda() for k, v in batch.items()}, method="test")["answer"]
            scores[str(dt)] = eval_rouge_l([" ".join(w) for w in model.to_sentence(None, got)], [[t] for t in truth])
            same[str(dt)] = float((got.cpu() == ans).all(dim=1).float().mean())
            prefix[str(dt)] = float(((got.cpu() == ans).long().cumprod(dim=1).sum(dim=1).float() / T).mean())  # common prefix / T
        finally:
            case_rg_amd.set_compute_dtype(torch.float32)
    record_error("rouge_l_synthetic_dev", "fp32", "rouge_l_points_vs_oracle", abs(scores["torch.float32"] - want), 0.2)
    record_error("rouge_l_synthetic_dev", "bf16_auto", "rouge_l_points_vs_oracle", abs(scores["torch.bfloat16"] - want), 0.2)
    record_error("rouge_l_synthetic_dev", "fp32", "fraction_of_answers_differing_from_oracle", 1.0 - same["torch.float32"], 0.03)
    record_error("rouge_l_synthetic_dev", "bf16_auto", "fraction_of_answers_differing_from_oracle", 1.0 - same["torch.bfloat16"], 0.5)
    record_error("rouge_l_synthetic_dev", "bf16_auto", "mean_common_prefix_fraction_short_of_1", 1.0 - prefix["torch.bfloat16"], 0.3)
    # asserted (round 3 only recorded them): fp32 decodes the oracle's answers; in bf16 one near-tie anywhere changes an answer from
    # there on (12 tokens, margins of a random model), so the bar is on whole answers AND on the prefix that agrees
    assert same["torch.float32"] >= 0.97, "fp32: %.3f of the answers equal the oracle's" % same["torch.float32"]
    assert same["torch.bfloat16"] >= 0.5, "bf16: %.3f of the answers equal the oracle's" % same["torch.bfloat16"]
    assert prefix["torch.bfloat16"] >= 0.7, "bf16: mean common prefix %.3f of T" % prefix["torch.bfloat16"]
    assert want > 0.0, "degenerate dev set"
    assert len(set(map(tuple, ans.tolist()))) > 8, "the oracle's answers collapsed to a few strings"
    assert abs(scores["torch.float32"] - want) <= 0.2, "fp32: ROUGE-L %.2f vs oracle %.2f" % (scores["torch.float32"], want)
    assert abs(scores["torch.bfloat16"] - want) <= 0.2, "bf16: ROUGE-L %.2f vs oracle %.2f" % (scores["torch.bfloat16"], want)
