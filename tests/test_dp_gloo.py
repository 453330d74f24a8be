"""The N > 1 path on CPU: world_size-2 ``gloo`` processes drive ``case_rg_amd.parallel.GradSync`` (the bucketed
all-reduce the trainer uses on RCCL) around the CPU oracle model.  DP-averaged gradients must equal the mean of
the per-shard gradients computed in one process, parameters must be broadcast from rank 0, and gradient
accumulation must reduce only on the boundary step."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _model():
    import oracle
    from case_rg_amd.utils import make_vocab
    v2i, i2v = make_vocab(150)
    return oracle.Masque(5, i2v, v2i, 32, enc_layers=1, dec_layers=1)


def _shard(rank):
    from case_rg_amd.utils import synth_batch
    return synth_batch(2, 2, 10, 6, 5, 150, seed=100 + rank, model="masque")


def _grads(model, batch):
    model.zero_grad()
    losses = model(dict(batch), method="train")
    sum(l.mean() for l in losses).backward()
    return {n: p.grad.clone() for n, p in model.named_parameters()}


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from case_rg_amd.parallel import GradSync
        from case_rg_amd.utils import fill_params
        model = fill_params(_model(), 40 + rank).train()  # ranks start different: broadcast must fix that
        sync = GradSync(model, bucket_mb=0.05)             # tiny buckets -> many buckets, async launches
        assert len(sync.buckets) > 3
        # accumulation micro-step: no communication, gradients stay local
        sync.no_sync(True)
        local = _grads(model, _shard(rank))
        sync.finish()
        assert all(torch.equal(local[n], p.grad) for n, p in model.named_parameters())
        # boundary step: reduce (grads accumulate on top of the micro-step, as CumulativeTrainer does)
        sync.no_sync(False)
        losses = model(dict(_shard(rank)), method="train")
        sum(l.mean() for l in losses).backward()
        sync.finish()
        rec = {"grads": {n: p.grad.clone() for n, p in model.named_parameters()},
               "params": {n: p.detach().clone() for n, p in model.named_parameters()}}
        # the reduced gradients ARE views of the flat buckets (nothing is copied back)
        assert all(p.grad.data_ptr() == v.data_ptr() for b in sync.buckets for (p, _, _), v in zip(b["items"], b["views"]))
        # ranks with DIFFERENT graphs: rank 1 trains the selection head only (Masque 'ps_train'), so the decoder's hooks never
        # fire there and its buckets are flushed by finish(); both ranks must still issue the same sequence of collectives
        model.zero_grad()
        losses = model(dict(_shard(rank)), method="train" if rank == 0 else "ps_train")
        sum(l.mean() for l in losses).backward()
        sync.finish()
        rec["grads_mixed"] = {n: p.grad.clone() for n, p in model.named_parameters()}
        # an abandoned step: backward ran (some collectives are in flight), finish() never did; abort() must leave a clean slate
        model.zero_grad()
        sum(l.mean() for l in model(dict(_shard(rank)), method="train")).backward()
        sync.abort()
        assert sync._next == 0 and all(b["work"] is None and b["pending"] == len(b["items"]) for b in sync.buckets)
        model.zero_grad()
        sum(l.mean() for l in model(dict(_shard(rank)), method="train")).backward()
        sync.finish()
        rec["grads_after_abort"] = {n: p.grad.clone() for n, p in model.named_parameters()}
        # the bf16 wire (CASE_DP_BF16 / comm_dtype): gradients rounded once, summed, divided -- a second GradSync on the same model
        for h in sync._handles:
            h.remove()
        wire = GradSync(model, bucket_mb=0.05, comm_dtype=torch.bfloat16)
        model.zero_grad()
        sum(l.mean() for l in model(dict(_shard(rank)), method="train")).backward()
        wire.finish()
        rec["grads_bf16_wire"] = {n: p.grad.clone() for n, p in model.named_parameters()}
        torch.save(rec, os.path.join(out_dir, "rank%d.pt" % rank))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_gradsync_world2_gloo(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "rank0.pt"), torch.load(tmp_path / "rank1.pt")
    from case_rg_amd.utils import fill_params
    ref = fill_params(_model(), 40).train()  # rank 0's parameters
    for n, p in ref.named_parameters():
        assert torch.equal(r0["params"][n], p) and torch.equal(r1["params"][n], p), "broadcast of " + n
    g0, g1 = _grads(ref, _shard(0)), _grads(ref, _shard(1))
    for n in g0:
        want = 2.0 * 0.5 * (g0[n] + g1[n])  # micro-step + boundary step, each averaged over 2 ranks
        assert torch.allclose(r0["grads"][n], want, rtol=1e-5, atol=1e-7), n
        assert torch.equal(r0["grads"][n], r1["grads"][n]), "ranks disagree on " + n
    ref.zero_grad()
    sum(l.mean() for l in ref(dict(_shard(1)), method="ps_train")).backward()
    for n, p in ref.named_parameters():
        g1 = torch.zeros_like(p) if p.grad is None else p.grad
        assert torch.allclose(r0["grads_mixed"][n], 0.5 * (g0[n] + g1), rtol=1e-5, atol=1e-7), "mixed graphs: " + n
        assert torch.equal(r0["grads_mixed"][n], r1["grads_mixed"][n]), "ranks disagree on " + n
    g1 = _grads(ref, _shard(1))
    for n in g0:
        want = 0.5 * (g0[n] + g1[n])
        assert torch.allclose(r0["grads_after_abort"][n], want, rtol=1e-5, atol=1e-7), "after abort: " + n
        assert torch.equal(r0["grads_after_abort"][n], r1["grads_after_abort"][n])
        # bf16 wire: each rank's gradient rounded to bf16 (2^-9 relative), the sum rounded again
        got = r0["grads_bf16_wire"][n]
        assert torch.equal(got, r1["grads_bf16_wire"][n]), "bf16 wire: ranks disagree on " + n
        assert (got - want).abs().max() <= 1.2e-2 * want.abs().max() + 1e-8, "bf16 wire: " + n


def _worker8(rank, world, port, out_dir):
    """Eight ranks (the node BASELINE.json's DP = 8 names): bucket order with eight participants, two ranks on another graph
    (Masque 'ps_train': the decoder's hooks never fire there), one abandoned step, then a clean step."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from case_rg_amd.parallel import GradSync
        from case_rg_amd.utils import fill_params
        model = fill_params(_model(), 40 + rank).train()
        sync = GradSync(model, bucket_mb=0.05)
        launched = []
        launch = sync._launch

        def watched(b):
            launched.append(next(i for i, x in enumerate(sync.buckets) if x is b))
            launch(b)

        sync._launch = watched
        rec = {"params": {n: p.detach().clone() for n, p in model.named_parameters()}, "nbuckets": len(sync.buckets)}
        # step 1: ranks 3 and 6 run the selection-only graph
        method = "ps_train" if rank in (3, 6) else "train"
        sum(l.mean() for l in model(dict(_shard(rank)), method=method)).backward()
        sync.finish()
        rec["order_mixed"] = list(launched)
        rec["grads_mixed"] = {n: p.grad.clone() for n, p in model.named_parameters()}
        # step 2: abandoned between backward and finish() on EVERY rank (collectives are in flight), then a clean step
        del launched[:]
        model.zero_grad()
        sum(l.mean() for l in model(dict(_shard(rank)), method="train")).backward()
        sync.abort()
        assert sync._next == 0 and all(b["work"] is None and b["pending"] == len(b["items"]) for b in sync.buckets)
        del launched[:]
        model.zero_grad()
        sum(l.mean() for l in model(dict(_shard(rank)), method="train")).backward()
        sync.finish()
        rec["order_clean"] = list(launched)
        rec["grads_clean"] = {n: p.grad.clone() for n, p in model.named_parameters()}
        torch.save(rec, os.path.join(out_dir, "rank%d.pt" % rank))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_gradsync_world8_gloo(tmp_path):
    """DP readiness without hardware: the bucket protocol at the world size of the target node.  Every rank issues
    the collectives in the same (index) order whatever its graph; the averaged gradients equal the mean of the eight shard gradients
    computed in one process (absent gradients count as zeros); an aborted step leaves no residue."""
    world = 8
    port = _free_port()
    mp.spawn(_worker8, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    recs = [torch.load(tmp_path / ("rank%d.pt" % r)) for r in range(world)]
    from case_rg_amd.utils import fill_params
    ref = fill_params(_model(), 40).train()
    nb = recs[0]["nbuckets"]
    assert nb > 3
    for r in recs:
        assert r["nbuckets"] == nb and r["order_mixed"] == list(range(nb)) and r["order_clean"] == list(range(nb)), "collectives must go out in bucket order on every rank"
        for n, p in ref.named_parameters():
            assert torch.equal(r["params"][n], p), "broadcast of " + n
    want_mixed = {n: torch.zeros_like(p) for n, p in ref.named_parameters()}
    want_clean = {n: torch.zeros_like(p) for n, p in ref.named_parameters()}
    for rank in range(world):
        ref.zero_grad()
        sum(l.mean() for l in ref(dict(_shard(rank)), method="ps_train" if rank in (3, 6) else "train")).backward()
        for n, p in ref.named_parameters():
            if p.grad is not None:
                want_mixed[n] += p.grad / world
        g = _grads(ref, _shard(rank))
        for n in g:
            want_clean[n] += g[n] / world
    for n in want_clean:
        for r in recs[1:]:
            assert torch.equal(r["grads_mixed"][n], recs[0]["grads_mixed"][n]) and torch.equal(r["grads_clean"][n], recs[0]["grads_clean"][n]), "ranks disagree on " + n
        assert torch.allclose(recs[0]["grads_mixed"][n], want_mixed[n], rtol=2e-5, atol=2e-7), "mixed graphs: " + n
        assert torch.allclose(recs[0]["grads_clean"][n], want_clean[n], rtol=2e-5, atol=2e-7), "after abort: " + n
