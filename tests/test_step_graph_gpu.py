"""Round 6: the device-resident step state (ABI 600, CaseStepState) and the hipGraph-captured training step
(case_rg_amd/stepstate.py, case_rg_amd/stepgraph.py; reference loop common/CumulativeTrainer.py:52-78, default geometry CaSE/Run.py:72-78).

What must hold for a captured step to be a drop-in for the eager one:
  * a dropout site given (offset, state) draws the mask of (offset + state.rng_base) -- at every kind of site (element-wise, GEMM
    epilogue, attention probabilities, embedding, LayerNorm-backward dual output);
  * the optimizer kernel with the device struct takes the same step, bit for bit, as with the per-tensor table scalars;
  * a replayed step draws NEW masks every time and the same masks as the eager step at that position of the counter stream;
  * N replayed steps follow N eager steps of a second trainer (same init, same batches, dropout ON) -- losses and parameters to the
    noise of the f32 atomics that make the eager step itself differ from run to run in the last bits;
  * host-side bookkeeping survives: per-parameter step counts, scheduler, RNG position, EMA swap for evaluation between replays."""
import ctypes as C
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture
def dropout_on():
    from case_rg_amd import config
    config.set_dropout(True)
    config.manual_seed(4321)
    yield
    config.set_device_state(None)
    config.set_dropout(False)
    config.set_compute_dtype(torch.float32)


def test_step_state_moves_every_kind_of_dropout_site(dropout_on):
    """(offset = 0, state.rng_base = B) == (offset = B, no state), bit for bit, at each site kind; and a different base, a different mask."""
    from case_rg_amd import config, ops
    from case_rg_amd.stepstate import StepState
    st = StepState(torch.device("cuda", 0))
    config.set_compute_dtype(torch.bfloat16)
    torch.manual_seed(0)
    x = torch.randn(512, 512, device=DEV, dtype=torch.bfloat16)
    w = (torch.randn(512, 512, device=DEV) * 0.05)
    qkv = torch.randn(4, 96, 3 * 512, device=DEV, dtype=torch.bfloat16) * 0.5
    ids = torch.randint(1, 300, (8, 64), device=DEV)
    table, pe = torch.randn(300, 512, device=DEV), torch.randn(64, 512, device=DEV)
    B = 123456  # even

    def sites():
        out = [ops.dropout(x, 0.3)]                                            # case_dropout
        out.append(ops.linear(x, w, None, None, 0.2))                          # GEMM epilogue (256 tiling)
        out.append(ops.linear(x[:192], w[:192], None, None, 0.2))              # GEMM epilogue (small tiling)
        out.append(ops.attention(qkv, qkv, qkv, 0, 512, 1024, 8, 64, p_drop=0.25))  # attention probabilities (fused, d = 64)
        out.append(ops.embed_pos(ids, table, pe, p_drop=0.1))                  # embedding
        out.append(ops.masked_softmax(x.float().reshape(8, 64, 512), p_drop=0.2, out_dtype=torch.bfloat16))  # softmax kernels
        return [t.float().clone() for t in out]

    config.set_device_state(None)
    config.set_rng_state((4321, B))
    want = sites()
    config.set_device_state(st.address)
    config.set_rng_state((4321, B))  # device mode: the position goes into the base, the sites are numbered from 0
    assert config.begin_step() == B
    st.upload(B)
    got = sites()
    for i, (a, b) in enumerate(zip(got, want)):
        assert torch.equal(a, b), "site %d: (0, base) differs from (base, no state)" % i
    config.set_rng_state((4321, B))  # the same site offsets again ...
    st.upload(B + 2)                 # ... on another base
    other = sites()
    assert all(not torch.equal(a, b) for a, b in zip(other, want)), "another base must draw other masks"
    assert st.read().rng_base == B + 2


def test_step_advance_kernel_equals_the_host_scalars():
    """case_step_advance forms lr / (1 - beta1^t) and sqrt(1 - beta2^t) in double on the device: equal (to one f32 ulp) to what the
    host writes into the table entries, for 1 .. 40 steps; rng_base moves by the stride; an odd stride is refused."""
    from case_rg_amd import _abi as A
    from case_rg_amd.stepstate import StepState
    st = StepState(torch.device("cuda", 0))
    st.host.lr, st.host.step = 2.5e-4, 0
    st.upload(1000)
    for t in range(1, 41):
        st.advance_on_device(4096, 0.9, 0.999)
        got = st.read()
        assert got.step == t and got.rng_base == 1000 + 4096 * t
        want_ss, want_bc = C.c_float(C.c_float(2.5e-4).value / (1.0 - 0.9 ** t)).value, C.c_float(math.sqrt(1.0 - 0.999 ** t)).value
        assert abs(got.step_size - want_ss) <= 1.2e-7 * want_ss and abs(got.bc2_sqrt - want_bc) <= 1.2e-7 * want_bc, (t, got.step_size, want_ss)
    with pytest.raises(RuntimeError, match="odd"):
        A.call("case_step_advance", st.address, 3, 0.9, 0.999, 0)


def test_fused_adam_reads_the_same_step_from_the_device_struct():
    """FusedAdam.step(state=...) against FusedAdam.step() on twin parameters: bit-identical parameters, moments and EMA shadows over four
    steps with a changing learning rate; a parameter whose step count differs falls back to the table scalars (and is still exact)."""
    from case_rg_amd.common.EMA import EMA
    from case_rg_amd.optim import FusedAdam
    from case_rg_amd.stepstate import StepState
    from test_optim_gpu import _Holder, _params
    dev = torch.device("cuda", 0)
    ma, mb = _Holder(_params(dev, 3)), _Holder(_params(dev, 3))
    ea, eb = EMA(ma, 0.995), EMA(mb, 0.995)
    ea.register(), eb.register()
    oa = FusedAdam(ma.parameters(), lr=2.5e-4, low_precision=torch.bfloat16)
    ob = FusedAdam(mb.parameters(), lr=2.5e-4, low_precision=torch.bfloat16)
    st = StepState(dev)
    g = torch.Generator().manual_seed(5)
    for step in range(4):
        for grp in oa.param_groups + ob.param_groups:
            grp["lr"] = 2.5e-4 * (1 + step)  # a scheduler
        skip = step == 2
        for i, (pa, pb) in enumerate(zip(ma.parameters(), mb.parameters())):
            gr = (torch.randn(pa.shape, generator=g) * 0.1).to(dev)
            pa.grad, pb.grad = (None, None) if (skip and i == 1) else (gr.clone(), gr.clone())
        oa.stage_step(st)
        st.upload(0)
        oa.step(clip_norm=1.0, ema=ea, state=st)
        ob.step(clip_norm=1.0, ema=eb)
        for (n, pa), pb in zip(ma.named_parameters(), mb.parameters()):
            assert torch.equal(pa, pb), (step, n)
            assert torch.equal(ea.shadow[n], eb.shadow[n])
            if pa in oa.state and "exp_avg" in oa.state[pa]:
                assert torch.equal(oa.state[pa]["exp_avg_sq"], ob.state[pb]["exp_avg_sq"]) and oa.state[pa]["step"] == ob.state[pb]["step"]


def _tiny_trainer(dtype, capture, seed=40):
    import case_rg_amd
    from case_rg_amd.CaSE.Model import CaSE
    from case_rg_amd.common.CumulativeTrainer import CumulativeTrainer
    from case_rg_amd.common.schedule import get_cosine_with_hard_restarts_schedule_with_warmup
    from case_rg_amd.optim import FusedAdam
    from case_rg_amd.utils import fill_params, make_vocab
    case_rg_amd.set_compute_dtype(dtype)
    case_rg_amd.set_dropout(True)
    case_rg_amd.config.manual_seed(99)
    v2i, i2v = make_vocab(300)
    model = fill_params(CaSE(4, 8, i2v, v2i, 64), seed).train()
    trainer = CumulativeTrainer(model, None, None, 0, 1, capture=capture)
    opt = FusedAdam(model.parameters(), lr=1e-3, low_precision=torch.bfloat16 if dtype == torch.bfloat16 else None)
    sched = get_cosine_with_hard_restarts_schedule_with_warmup(opt, 3, 100)
    return trainer, opt, sched


def _batch(step):
    from case_rg_amd.utils import synth_batch
    return {k: v.cuda() for k, v in synth_batch(2, 3, 24, 12, 8, 300, seed=500 + step, ragged=False, model="case").items()}


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_replayed_steps_follow_the_eager_steps(dropout_on, dtype):
    """Two trainers, same initial weights, same six batches, dropout ON: one runs every step eagerly (with the device-resident state, so
    that both number their dropout sites alike), the other runs two eager steps, records the third and replays it four times.  Losses,
    parameters, EMA shadows, Adam step counts, the LR schedule and the position of the dropout counter stream must agree."""
    from case_rg_amd import config
    runs = []
    for capture_steps in (False, True):
        trainer, opt, sched = _tiny_trainer(dtype, capture=True)
        if not capture_steps:
            trainer.graphs = None  # device-state mode without captures: the eager yardstick
        losses = [trainer.train_batch(0, _batch(s), "train", opt, sched) for s in range(6)]
        torch.cuda.synchronize()
        runs.append(dict(losses=losses, params={n: p.detach().clone() for n, p in trainer.model.named_parameters()},
                         ema={n: t.clone() for n, t in trainer.ema.shadow.items()}, steps=sorted({int(s["step"]) for s in opt.state.values()}),
                         lr=sched.get_last_lr(), rng=config.rng_state(), replays=0 if trainer.graphs is None else trainer.graphs.replays))
        trainer.close()
    eager, graph = runs
    assert graph["replays"] == 4 and eager["replays"] == 0
    assert graph["steps"] == eager["steps"] == [6] and graph["lr"] == eager["lr"] and graph["rng"] == eager["rng"]
    # the weight / LayerNorm / embedding gradients accumulate with f32 atomics: two EAGER runs already differ in the last bits
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    for s, (a, b) in enumerate(zip(eager["losses"], graph["losses"])):
        assert all(abs(x - y) <= tol * max(1.0, abs(x)) for x, y in zip(a, b)), (s, a, b)
    assert eager["losses"][3] != eager["losses"][4]
    worst = 0.0
    for n, want in eager["params"].items():
        err = (graph["params"][n] - want).abs().max().item() / (want.abs().max().item() + 1e-12)
        worst = max(worst, err)
        assert err <= (5e-3 if dtype == torch.float32 else 5e-2), (n, err)
        assert (graph["ema"][n] - eager["ema"][n]).abs().max().item() <= (5e-3 if dtype == torch.float32 else 5e-2) * (want.abs().max().item() + 1e-12)
    print("captured vs eager steps (%s): worst parameter deviation %.2e" % (dtype, worst))


def test_replays_draw_new_masks_and_the_eager_masks(dropout_on):
    """The masks themselves: a captured step whose 'model' is one dropout site.  Replay k draws the mask the eager call draws at the
    same position of the counter stream, and consecutive replays differ."""
    from case_rg_amd import config, ops
    from case_rg_amd.stepstate import StepState
    st = StepState(torch.device("cuda", 0))
    config.set_device_state(st.address)
    x = torch.ones(256, 512, device=DEV, dtype=torch.bfloat16)
    eager = []
    for k in range(3):
        st.upload(config.begin_step())
        eager.append(ops.dropout(x, 0.5).clone())
    config.manual_seed(4321)
    config.set_device_state(st.address)
    st.upload(config.begin_step())
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        y = ops.dropout(x, 0.5)
    consumed = config.rng_state()[1]
    config.skip_rng(-consumed)
    got = []
    for k in range(3):
        st.upload(config.begin_step())
        graph.replay()
        config.skip_rng(consumed)
        got.append(y.clone())
    torch.cuda.synchronize()
    for k in range(3):
        assert torch.equal(got[k], eager[k]), "replay %d drew another mask than the eager step at the same stream position" % k
    assert not torch.equal(got[0], got[1]) and not torch.equal(got[1], got[2])
    assert 0.45 < (got[2] == 0).float().mean().item() < 0.55


def test_evaluation_between_replays_and_checkpoint_reset(dropout_on, tmp_path):
    """EMA.apply_shadow / restore (evaluation on the averaged weights) between two replays: parameters are swapped through .data and the
    operand cache is dropped -- the next replay must train on the restored weights, like the eager loop.  A checkpoint load replaces the
    EMA shadows and the optimizer moments: the captures are dropped and the loop continues (eagerly, then re-captured).  The same
    scenario runs twice, one after the other (the dropout counter stream is process-wide): eager yardstick, then captured."""
    def scenario(captured):
        trainer, opt, sched = _tiny_trainer(torch.bfloat16, capture=True)
        if not captured:
            trainer.graphs = None
        out = [trainer.train_batch(0, _batch(s), "train", opt, sched) for s in range(4)]
        assert not captured or trainer.graphs.replays == 2
        trainer.ema.apply_shadow()
        trainer.model.eval()
        with torch.no_grad():
            trainer.model(_batch(50), method="test")
        trainer.model.train()
        trainer.ema.restore()
        out.append(trainer.train_batch(0, _batch(4), "train", opt, sched))
        assert not captured or trainer.graphs.replays == 3
        path = trainer.save_checkpoint(0, str(tmp_path), opt, sched)
        trainer.load_checkpoint(path, opt, sched)
        assert not captured or not trainer.graphs.graphs
        out += [trainer.train_batch(0, _batch(s), "train", opt, sched) for s in range(5, 9)]
        assert not captured or trainer.graphs.replays == 5  # two eager steps, the recording + its replay, one more replay
        trainer.close()
        return out

    eager, graph = scenario(False), scenario(True)
    for s, (a, b) in enumerate(zip(eager, graph)):
        assert all(math.isfinite(x) for x in b)
        assert all(abs(x - y) <= 3e-2 * max(1.0, abs(x)) for x, y in zip(a, b)), (s, a, b)


def test_a_step_that_cannot_be_captured_stays_eager(dropout_on):
    """A forward that reads a device value on the host (here: a hook calling .item()) cannot be recorded: the capture attempt is abandoned with
    a warning, its host-side effects are taken back (step counts, RNG position) and that batch shape keeps running eagerly -- same losses as a
    trainer that never tried."""
    import warnings

    def hook(module, args, output):
        float(output[0].sum().item())  # a host read: illegal while the stream is capturing

    results = []
    for capture in (False, True):
        trainer, opt, sched = _tiny_trainer(torch.float32, capture=True)
        handle = trainer.model.register_forward_hook(hook)
        if not capture:
            trainer.graphs = None
        with warnings.catch_warnings(record=True) as seen:
            warnings.simplefilter("always")
            losses = [trainer.train_batch(0, _batch(s), "train", opt, sched) for s in range(5)]
        handle.remove()
        if capture:
            assert trainer.graphs.replays == 0 and len(trainer.graphs.disabled) == 1
            assert any("could not be captured" in str(w.message) for w in seen)
        results.append((losses, sorted({int(s["step"]) for s in opt.state.values()})))
        trainer.close()
    (a, sa), (b, sb) = results
    assert sa == sb == [5]
    for x, y in zip(a, b):
        assert all(abs(u - v) <= 2e-5 * max(1.0, abs(u)) for u, v in zip(x, y)), (x, y)


def test_ragged_answers_share_one_captured_step():
    """The reference's collate pads data['response'] to the batch's longest answer (CaSE/CaSEDataset.py:135-136): the capturing trainer pads
    it further to the model's max_target_length, so batches with answers of 5 .. 8 tokens replay ONE graph; PAD targets change neither the
    losses nor the training trajectory (fp32, dropout off: against an eager trainer that sees the unpadded batches)."""
    import case_rg_amd
    from case_rg_amd import config
    from case_rg_amd.utils import synth_batch

    def batch(step):
        return {k: v.cuda() for k, v in synth_batch(2, 3, 24, 12, 8, 300, seed=900 + step, ragged=True, model="case").items()}

    lens = {batch(s)["response"].shape[1] for s in range(7)}
    assert len(lens) >= 2, "the synthetic batches must differ in answer length for this test to mean anything"
    runs = []
    for capture in (False, True):
        trainer, opt, sched = _tiny_trainer(torch.float32, capture=capture)
        case_rg_amd.set_dropout(False)
        losses = [trainer.train_batch(0, batch(s), "train", opt, sched) for s in range(7)]
        if capture:
            assert trainer.graphs.replays == 5 and len(trainer.graphs.graphs) == 1
        runs.append((losses, {n: p.detach().clone() for n, p in trainer.model.named_parameters()}))
        trainer.close()
        config.set_device_state(None)
    (la, pa), (lb, pb) = runs
    for s, (x, y) in enumerate(zip(la, lb)):
        assert all(abs(u - v) <= 5e-5 * max(1.0, abs(u)) for u, v in zip(x, y)), (s, x, y)
    worst = max(((pb[n] - pa[n]).abs().max() / (pa[n].abs().max() + 1e-12)).item() for n in pa)
    assert worst <= 5e-3, worst


@pytest.mark.parametrize("eager_ms,kept", [(1e-3, False), (1e6, True)])
def test_auto_capture_keeps_a_graph_only_where_it_pays(dropout_on, eager_ms, kept):
    """capture="auto": the first three replays of a shape are timed against its eager warm-up steps; a capture that is not 3 % faster is
    dropped and the shape stays eager (the yardstick is planted here: a 1-ns eager step can never be beaten, a 1000-s one always)."""
    trainer, opt, sched = _tiny_trainer(torch.float32, capture="auto")
    assert trainer.graphs.auto
    losses = []
    for s in range(8):
        losses.append(trainer.train_batch(0, _batch(s), "train", opt, sched))
        for sig in trainer.graphs.eager_ms:
            trainer.graphs.eager_ms[sig] = [eager_ms]
    assert all(math.isfinite(x) for l in losses for x in l)
    assert sorted({int(s_["step"]) for s_ in opt.state.values()}) == [8]
    if kept:
        assert len(trainer.graphs.graphs) == 1 and trainer.graphs.replays == 6 and not trainer.graphs.disabled
    else:
        assert not trainer.graphs.graphs and trainer.graphs.replays == 3 and len(trainer.graphs.disabled) == 1
    trainer.close()
