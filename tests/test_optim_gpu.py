"""K15 / SURVEY f4 on the GPU: the multi-tensor clip + Adam + EMA + operand-refresh pass against the reference's three
separate calls (common/CumulativeTrainer.py:70-76: clip_grad_norm_(params, 1); optimizer.step() with CaSE/Run.py:27's
optim.Adam; common/EMA.py:13-18) executed by torch on the same tensors."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _params(dev, seed):
    g = torch.Generator().manual_seed(seed)
    shapes = [(7,), (33, 17), (1,), (128, 512), (16385,), (3, 5, 7), (40000, 3)]  # odd sizes, > 1 chunk, unaligned tails
    return [torch.nn.Parameter(torch.randn(s, generator=g).to(dev)) for s in shapes]


class _Holder(torch.nn.Module):
    def __init__(self, ps):
        super().__init__()
        self.ps = torch.nn.ParameterList(ps)


@pytest.mark.parametrize("clip", [None, 1.0, 1e4])
def test_fused_adam_matches_torch_clip_adam_ema(clip):
    from case_rg_amd.common.EMA import EMA
    from case_rg_amd.optim import FusedAdam
    dev = torch.device("cuda", 0)
    ma, mb = _Holder(_params(dev, 3)), _Holder(_params(dev, 3))
    ea, eb = EMA(ma, 0.995), EMA(mb, 0.995)
    ea.register(), eb.register()
    oa = FusedAdam(ma.parameters(), lr=2.5e-4, low_precision=torch.bfloat16)
    ob = torch.optim.Adam(mb.parameters(), lr=2.5e-4)
    g = torch.Generator().manual_seed(5)
    for step in range(4):
        scale = 10.0 if step == 1 else 0.01  # one step above the clip threshold, the others below
        for pa, pb in zip(ma.parameters(), mb.parameters()):
            gr = (torch.randn(pa.shape, generator=g) * scale).to(dev)
            pa.grad, pb.grad = gr.clone(), gr.clone()
        oa.step(clip_norm=clip, ema=ea)
        if clip is not None:
            torch.nn.utils.clip_grad_norm_(mb.parameters(), clip)
        ob.step()
        eb.update()
        for (n, pa), pb in zip(ma.named_parameters(), mb.parameters()):
            assert torch.allclose(pa, pb, rtol=1e-5, atol=1e-7), (step, n, (pa - pb).abs().max().item())
            assert torch.allclose(ea.shadow[n], eb.shadow[n], rtol=1e-6, atol=1e-7), (step, n)
            sa, sb = oa.state[pa], ob.state[pb]
            assert torch.allclose(sa["exp_avg"], sb["exp_avg"], rtol=1e-5, atol=1e-9)
            assert torch.allclose(sa["exp_avg_sq"], sb["exp_avg_sq"], rtol=1e-5, atol=1e-12)
            assert sa["step"] == int(sb["step"])
    # the state dict is torch's layout: a torch.optim.Adam loads it and continues
    oc = torch.optim.Adam(ma.parameters(), lr=2.5e-4)
    oc.load_state_dict(oa.state_dict())
    assert int(oc.state[next(iter(ma.parameters()))]["step"]) == 4


def test_fused_adam_refreshes_the_bf16_operand_cache():
    from case_rg_amd import ops
    from case_rg_amd.optim import FusedAdam
    dev = torch.device("cuda", 0)
    w = torch.nn.Parameter(torch.randn(96, 64, device=dev))
    b = torch.nn.Parameter(torch.randn(64, device=dev))
    opt = FusedAdam([w, b], lr=0.1, low_precision=torch.bfloat16)
    stale = ops.cast_param(w, torch.bfloat16).clone()
    w.grad, b.grad = torch.ones_like(w), torch.ones_like(b)
    opt.step()
    low = ops.cast_param(w, torch.bfloat16)
    assert torch.equal(low, w.detach().to(torch.bfloat16)) and not torch.equal(low, stale)
    rows = ops.cast_param(w[32:64], torch.bfloat16)  # a view of the parameter (K rows of an in_proj_weight) slices the same copy
    assert rows.data_ptr() == low.data_ptr() + 32 * 64 * 2 and torch.equal(rows, w.detach()[32:64].to(torch.bfloat16))
    before = low.data_ptr()
    w.grad, b.grad = torch.ones_like(w), torch.ones_like(b)
    opt.step()
    low2 = ops.cast_param(w, torch.bfloat16)
    assert low2.data_ptr() == before and torch.equal(low2, w.detach().to(torch.bfloat16)), "refreshed in place, no new allocation"
    with torch.no_grad():
        w.mul_(2.0)  # an in-place update outside the optimizer bumps _version: the seeded copy must not be served
    assert torch.equal(ops.cast_param(w, torch.bfloat16), w.detach().to(torch.bfloat16))
    cpu = torch.nn.Parameter(torch.zeros(3))
    cpu.grad = torch.ones(3)
    with pytest.raises(RuntimeError, match="no CPU path"):
        FusedAdam([cpu], lr=0.1).step()


def test_trainer_with_fused_adam_follows_the_unfused_loop():
    """CumulativeTrainer recognises FusedAdam: same losses / weights / EMA as clip_grad_norm_ + Adam + EMA.update (fp32, dropout off)."""
    import case_rg_amd
    from case_rg_amd.CaSE.Model import CaSE
    from case_rg_amd.common.CumulativeTrainer import CumulativeTrainer
    from case_rg_amd.optim import FusedAdam
    from case_rg_amd.utils import fill_params, make_vocab, synth_batch
    case_rg_amd.set_compute_dtype(torch.float32)
    case_rg_amd.set_dropout(False)
    dev = torch.device("cuda", 0)
    v2i, i2v = make_vocab(200)
    batches = [{k: v.to(dev) for k, v in synth_batch(2, 3, 16, 8, 6, 200, seed=20 + i, model="case").items()} for i in range(5)]
    traj = []
    for fused in (True, False):
        model = fill_params(CaSE(4, 6, i2v, v2i, 64, enc_layers=1), 4).to(dev).train()
        tr = CumulativeTrainer(model, None, None, None, 1, accumulation_steps=2)
        opt = FusedAdam(model.parameters(), lr=1e-3) if fused else torch.optim.Adam(model.parameters(), lr=1e-3)
        losses = [tr.train_batch(0, dict(b), "train", opt) for b in batches]
        traj.append((losses, {n: p.detach().clone() for n, p in model.named_parameters()}, dict(tr.ema.shadow)))
    (la, pa, ea), (lb, pb, eb) = traj
    assert torch.allclose(torch.tensor(la), torch.tensor(lb), rtol=1e-4, atol=1e-5)
    for n in pa:
        assert torch.allclose(pa[n], pb[n], rtol=1e-3, atol=2e-5), (n, (pa[n] - pb[n]).abs().max().item())
        assert torch.allclose(ea[n], eb[n], rtol=1e-4, atol=1e-6), n


def test_fused_adam_keeps_a_step_per_parameter_and_moves_the_shadow_of_idle_ones():
    """A parameter whose gradient is None on some steps (Masque alternating 'ps_train' / 'train', data-dependent branches): torch.optim.Adam
    skips it and keeps its own step count; the reference's EMA.update() still lerps its shadow (common/EMA.py:13-18)."""
    from case_rg_amd.common.EMA import EMA
    from case_rg_amd.optim import FusedAdam
    dev = torch.device("cuda", 0)
    ma, mb = _Holder(_params(dev, 9)), _Holder(_params(dev, 9))
    ea, eb = EMA(ma, 0.9), EMA(mb, 0.9)
    ea.register(), eb.register()
    oa = FusedAdam(ma.parameters(), lr=1e-2, low_precision=torch.bfloat16)
    ob = torch.optim.Adam(mb.parameters(), lr=1e-2)
    g = torch.Generator().manual_seed(6)
    for step in range(5):
        for i, (pa, pb) in enumerate(zip(ma.parameters(), mb.parameters())):
            gr = torch.randn(pa.shape, generator=g).to(dev) * 0.1
            idle = (i in (1, 3) and step in (0, 2)) or (i == 4 and step < 3)  # late first gradient, gaps
            pa.grad, pb.grad = (None, None) if idle else (gr.clone(), gr.clone())
        oa.step(clip_norm=1.0, ema=ea)
        torch.nn.utils.clip_grad_norm_(mb.parameters(), 1.0)
        ob.step()
        eb.update()
        for (n, pa), pb in zip(ma.named_parameters(), mb.parameters()):
            assert torch.allclose(pa, pb, rtol=1e-5, atol=1e-7), (step, n, (pa - pb).abs().max().item())
            assert torch.allclose(ea.shadow[n], eb.shadow[n], rtol=1e-6, atol=1e-7), (step, n)
            if pb in ob.state and ob.state[pb]:
                assert oa.state[pa]["step"] == int(ob.state[pb]["step"]), (step, n)


def test_fused_adam_clip_is_bit_reproducible():
    """The squared gradient norm is a fixed-order sum (per-chunk partials, then one workgroup): the clip coefficient -- and so every
    updated parameter -- is bit-identical run to run and across data-parallel ranks holding the same reduced gradients, whatever else
    the GPU is doing (f32 atomics into one scalar were not)."""
    from case_rg_amd.optim import FusedAdam
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(12)
    shapes = [(30522, 64), (2560, 2560), (777,), (512, 512), (3,)]
    init = [torch.randn(s, generator=g).to(dev) for s in shapes]
    grads = [(torch.randn(s, generator=g) * 3.0).to(dev) for s in shapes]
    side = torch.cuda.Stream()
    noise = torch.randn(4096, 4096, device=dev)
    results = []
    for run in range(4):
        ps = [torch.nn.Parameter(t.clone()) for t in init]
        opt = FusedAdam(ps, lr=1e-2)
        if run % 2:  # perturb workgroup scheduling with a concurrent kernel stream
            with torch.cuda.stream(side):
                for _ in range(4):
                    noise = noise @ noise * 1e-4
        for _ in range(3):
            for p, gr in zip(ps, grads):
                p.grad = gr.clone()
            opt.step(clip_norm=1.0)
        torch.cuda.synchronize()
        results.append(([p.detach().clone() for p in ps], opt._norm_ws[0].item()))
    for ps, nrm in results[1:]:
        assert nrm == results[0][1]
        assert all(torch.equal(a, b) for a, b in zip(ps, results[0][0]))
    want = sum(float((gr.double() ** 2).sum()) for gr in grads)
    assert abs(results[0][1] - want) <= 1e-5 * want
