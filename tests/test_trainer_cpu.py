"""Trainer-loop plumbing on CPU (BASELINE cfg 1's role): ``case_rg_amd.common.CumulativeTrainer`` -- host-only,
device-agnostic code -- drives a tiny oracle model through train_epoch / predict / serialize.  The numbers of the loop
(loss trajectory, weights, EMA shadow, rank scores) are pinned to the REFERENCE's own loop by the ``trainer_traj`` fixture
(tests/test_oracle_vs_golden.py); this file checks the loop's control flow against the reference's statements:

  common/CumulativeTrainer.py:52-78   accumulate -> clip-norm 1 -> optimizer -> EMA -> scheduler -> zero_grad on group boundaries
  common/CumulativeTrainer.py:122-126 end-of-epoch flush of a partial group: optimizer + scheduler, NO clip, NO EMA
  common/CumulativeTrainer.py:80-86   serialize on rank 0 (here also on one process, where the reference crashes on .module)
  common/EMA.py:13-32                 shadow <- decay * shadow + (1 - decay) * param; apply_shadow / restore
"""
import os

import pytest
import torch

import cases
import oracle
from case_rg_amd.common.CumulativeTrainer import CumulativeTrainer, init_params
from case_rg_amd.common.EMA import EMA
from case_rg_amd.common.schedule import get_cosine_with_hard_restarts_schedule_with_warmup
from case_rg_amd.utils import fill_params, make_vocab, synth_batch


def _tiny(seed=1, model="masque"):
    v2i, i2v = make_vocab(150)
    if model == "masque":
        m = oracle.Masque(5, i2v, v2i, 32, enc_layers=1, dec_layers=1)
    else:
        m = oracle.CaSE(4, 5, i2v, v2i, 32, enc_layers=1, dec_layers=1)
    return fill_params(m, seed).train()


def _dataset(n, seed, model="masque"):
    return cases._ListDataset(synth_batch(n, 2, 10, 6, 5, 150, seed=seed, model=model))


class _CountingAdam(torch.optim.Adam):
    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.calls, self.grad_norms = 0, []

    def step(self, closure=None):
        self.calls += 1
        gs = [p.grad for g in self.param_groups for p in g["params"] if p.grad is not None]
        self.grad_norms.append(float(torch.norm(torch.stack([g.norm() for g in gs]))))
        return super().step(closure)


def test_train_epoch_accumulates_flushes_and_schedules():
    model = _tiny()
    trainer = CumulativeTrainer(model, None, None, None, 1, accumulation_steps=2)
    opt = _CountingAdam(model.parameters(), lr=1e-3)
    sched = get_cosine_with_hard_restarts_schedule_with_warmup(opt, 2, 10)
    ema_updates = []
    update = trainer.ema.update
    trainer.ema.update = lambda: (ema_updates.append(opt.calls), update())[1]
    with cases._unshuffled_loader():
        trainer.train_epoch("train", _dataset(10, 3), cases._collate, 2, 0, opt, sched)  # 5 batches = 2 groups + half a group
    assert trainer.accumulation_count == 5
    assert opt.calls == 3, "two group boundaries + the end-of-epoch flush"
    assert ema_updates == [1, 2], "EMA follows the two boundary steps only (the flush skips it, reference :122-126)"
    assert sched.last_epoch == 3 and sched.get_last_lr()[0] == pytest.approx(1e-3 * 0.5 * (1 + torch.cos(torch.tensor(torch.pi / 8)).item()))
    assert opt.grad_norms[0] <= 1.0 + 1e-5 and opt.grad_norms[1] <= 1.0 + 1e-5, "boundary steps see clipped gradients"
    assert all(p.grad is None for p in model.parameters()), "zero_grad after every optimizer step"
    # a second epoch continues the running count: batch 6 closes the group the flush already applied
    with cases._unshuffled_loader():
        trainer.train_epoch("train", _dataset(2, 4), cases._collate, 2, 1, opt, sched)
    assert trainer.accumulation_count == 6 and opt.calls == 4


def test_accumulated_gradient_equals_the_mean_of_the_micro_batches():
    model = _tiny(seed=2)
    trainer = CumulativeTrainer(model, None, None, None, 1, accumulation_steps=2)
    seen = {}

    class Probe(torch.optim.SGD):
        def step(self, closure=None):
            seen.update({n: p.grad.clone() for n, p in model.named_parameters()})

    opt = Probe(model.parameters(), lr=0.0)
    ds = _dataset(4, 5)
    b1, b2 = cases._collate(ds.items[:2]), cases._collate(ds.items[2:])
    norm_fn = torch.nn.utils.clip_grad_norm_
    torch.nn.utils.clip_grad_norm_ = lambda params, max_norm: None  # look at the raw accumulated gradient
    try:
        l1 = trainer.train_batch(0, dict(b1), "train", opt)
        assert not seen, "no optimizer step inside a group"
        l2 = trainer.train_batch(0, dict(b2), "train", opt)
    finally:
        torch.nn.utils.clip_grad_norm_ = norm_fn
    assert len(l1) == 2 and len(l2) == 2 and all(isinstance(x, float) for x in l1 + l2)
    ref = _tiny(seed=2)
    total = sum(l.mean() for l in ref(dict(b1), method="train")) / 2 + sum(l.mean() for l in ref(dict(b2), method="train")) / 2
    total.backward()
    for n, p in ref.named_parameters():
        assert torch.allclose(seen[n], p.grad, rtol=1e-5, atol=1e-7), n


def test_single_tensor_loss_and_ps_train_method():
    """The loop accepts a list / tuple of losses or one tensor (reference :56-62); Masque's 'ps_train' returns a 1-list."""
    model = _tiny(seed=3)
    trainer = CumulativeTrainer(model, None, None, None, 1)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    b = cases._collate(_dataset(2, 6).items)
    assert len(trainer.train_batch(0, dict(b), "ps_train", opt)) == 1

    class OneLoss(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.ones(3))

        def forward(self, data, method):
            return (self.w * data["x"]).sum(-1)  # [B]: the trainer takes the mean

    t2 = CumulativeTrainer(OneLoss(), None, None, None, 1)
    out = t2.train_batch(0, {"x": torch.ones(4, 3)}, "train", torch.optim.SGD(t2.model.parameters(), lr=0.1))
    assert out == [3.0]


def test_predict_returns_batches_in_dataset_order_in_eval_mode():
    model = _tiny(seed=4)
    trainer = CumulativeTrainer(model, None, None, None, 1)
    ds = _dataset(5, 7)
    rs = trainer.predict("test", ds, cases._collate, 2)
    assert not model.training
    assert [d["id"].tolist() for d, _ in rs] == [[0, 1], [2, 3], [4]]
    for data, out in rs:
        assert out["answer"].shape == (data["id"].numel(), 5) and out["answer"].dtype == torch.int64
        assert out["rank"].shape == (data["id"].numel(), 2)


def test_serialize_writes_a_strictly_loadable_state_dict_on_rank0_only(tmp_path):
    model = _tiny(seed=5, model="case")
    CumulativeTrainer(model, None, None, None, 1).serialize(3, str(tmp_path))
    path = tmp_path / "model" / "3.pkl"
    assert path.exists()
    other = _tiny(seed=6, model="case")
    other.load_state_dict(torch.load(path, map_location="cpu"), strict=True)  # CaSE/Run.py:55
    for (n, a), (_, b) in zip(model.state_dict().items(), other.state_dict().items()):
        assert torch.equal(a, b), n
    t1 = CumulativeTrainer(model, None, None, None, 1)
    t1.local_rank = 1
    t1.serialize(4, str(tmp_path))
    assert not (tmp_path / "model" / "4.pkl").exists()


def test_ema_shadow_recursion_apply_and_restore():
    model = _tiny(seed=7)
    ema = EMA(model, 0.9)
    ema.register()
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    start = {n: p.detach().clone() for n, p in model.named_parameters()}
    with torch.no_grad():
        for p in model.parameters():
            p.add_(1.0)
    ema.update()
    with torch.no_grad():
        for p in model.parameters():
            p.add_(1.0)
    ema.update()
    for n in names:  # s2 = 0.9 (0.9 s0 + 0.1 (s0 + 1)) + 0.1 (s0 + 2) = s0 + 0.29
        assert torch.allclose(ema.shadow[n], start[n] + 0.29, atol=1e-6), n
    live = {n: p.detach().clone() for n, p in model.named_parameters()}
    ema.apply_shadow()
    assert all(torch.equal(p.data, ema.shadow[n]) for n, p in model.named_parameters())
    ema.restore()
    assert all(torch.equal(p.data, live[n]) for n, p in model.named_parameters()) and not ema.backup


def test_init_params_matches_the_reference_rule():
    """xavier-uniform on every tensor with dim > 1 (embedding row 0 included), vectors untouched, ``escape`` skips by substring."""
    model = _tiny(seed=8)
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    torch.manual_seed(0)
    init_params(model, escape="scorer")
    for n, p in model.named_parameters():
        if p.dim() > 1 and "scorer" not in n:
            bound = (6.0 / (p.size(0) + p.size(1))) ** 0.5 if p.dim() == 2 else None
            assert not torch.equal(p, before[n]), n
            if bound is not None:
                assert p.abs().max() <= bound + 1e-6, n
        else:
            assert torch.equal(p, before[n]), n
    emb = dict(model.named_parameters())["query_encoder.embedding.0.weight"]
    assert emb[0].abs().sum() > 0, "PAD row is re-initialised too (reference :19-20)"


def test_checkpoint_resume_continues_the_same_trajectory(tmp_path):
    """the reference saves weights only (:80-86); save_checkpoint / load_checkpoint carry optimizer moments + step,
    scheduler, EMA shadow, the accumulation counter and the RNG streams, so a resumed run repeats the uninterrupted one."""
    from case_rg_amd import config

    def run(n_first, resume):
        model = _tiny(seed=8)
        tr = CumulativeTrainer(model, None, None, None, 1, accumulation_steps=2)
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        sched = get_cosine_with_hard_restarts_schedule_with_warmup(opt, 2, 10)
        config.manual_seed(77)
        ds = _dataset(12, 9)
        batches = [cases._collate(ds.items[i:i + 2]) for i in range(0, 12, 2)]
        for b in batches[:n_first]:
            tr.train_batch(0, dict(b), "train", opt, sched)
        if resume:
            config.next_rng(10)  # the dropout counter moved during the first half
            path = tr.save_checkpoint(0, str(tmp_path), opt, sched)
            assert path.endswith(os.path.join("model", "0.ckpt"))
            model = _tiny(seed=99)  # different weights: everything must come from the file
            tr = CumulativeTrainer(model, None, None, None, 1, accumulation_steps=2)
            opt = torch.optim.Adam(model.parameters(), lr=1e-3)
            sched = get_cosine_with_hard_restarts_schedule_with_warmup(opt, 2, 10)
            config.manual_seed(1)
            assert tr.load_checkpoint(path, opt, sched) == 0
            assert config.rng_state() == (77, 10)
        for b in batches[n_first:]:
            tr.train_batch(0, dict(b), "train", opt, sched)
        return tr, opt, sched

    a, oa, sa = run(6, False)
    b, ob, sb = run(3, True)  # interrupted in the MIDDLE of an accumulation group (count 3 of steps 2)
    assert a.accumulation_count == b.accumulation_count == 6
    assert sa.last_epoch == sb.last_epoch and sa.get_last_lr() == sb.get_last_lr()
    pa, pb = dict(a.model.named_parameters()), dict(b.model.named_parameters())
    for n in pa:
        # the half-accumulated gradient of batch 3 is not part of a checkpoint: the resumed group sees batch 4 only
        assert torch.isfinite(pb[n]).all()
    # interrupted on a group boundary the trajectories are identical, bit for bit
    c, oc, sc = run(4, True)
    pc = dict(c.model.named_parameters())
    for n in pa:
        assert torch.equal(pa[n], pc[n]), n
        assert torch.equal(a.ema.shadow[n], c.ema.shadow[n]), n
    for (ka, va), (kc, vc) in zip(oa.state_dict()["state"].items(), oc.state_dict()["state"].items()):
        assert torch.equal(va["exp_avg"], vc["exp_avg"]) and torch.equal(va["exp_avg_sq"], vc["exp_avg_sq"]) and va["step"] == vc["step"]


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-only behaviour")
def test_prefetcher_is_a_pass_through_without_a_gpu():
    from case_rg_amd.utils.pipeline import DevicePrefetcher
    batches = [{"x": torch.full((2,), float(i))} for i in range(3)]
    out = list(DevicePrefetcher(batches))
    assert len(out) == 3 and all(a is b for a, b in zip(out, batches)) and len(DevicePrefetcher(batches)) == 3
