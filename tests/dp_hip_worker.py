"""One rank of tests/test_dp_hip_gpu.py: a FRESH process (its own HIP context on device 0) that trains the HIP CaSE model for three
steps through CumulativeTrainer + FusedAdam + GradSync over a gloo group (RCCL refuses two ranks on one device; the bucket protocol,
the bucket-view gradients the fused optimizer reads, and the parameter cache are the real ones).
usage: python tests/dp_hip_worker.py RANK WORLD PORT OUT_DIR [graph]   (graph: hipGraph-captured steps, two segments around the all-reduce)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build(seed):
    import torch
    import case_rg_amd
    from case_rg_amd.CaSE.Model import CaSE
    from case_rg_amd.utils import fill_params, make_vocab
    case_rg_amd.set_compute_dtype(torch.float32)
    case_rg_amd.set_dropout(False)
    v2i, i2v = make_vocab(300)
    return fill_params(CaSE(4, 8, i2v, v2i, 64), seed).train()


def shard(step, rank):
    from case_rg_amd.utils import synth_batch
    return synth_batch(2, 3, 24, 12, 8, 300, seed=100 + 10 * step + rank, ragged=False, model="case")


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from case_rg_amd import _abi
        from case_rg_amd.common.CumulativeTrainer import CumulativeTrainer
        from case_rg_amd.optim import FusedAdam
        model = build(40 + rank)  # ranks start DIFFERENT: the broadcast at GradSync construction must make them rank 0's
        graph = len(sys.argv) > 5 and sys.argv[5] == "graph"
        nsteps = 6 if graph else 3
        trainer = CumulativeTrainer(model, None, None, 0, world, capture=graph)
        assert trainer.sync is not None and trainer.sync.active and len(trainer.sync.buckets) >= 1
        opt = FusedAdam(model.parameters(), lr=1e-3)
        losses = []
        seen = []  # the library's reservation right behind every bucket launch (collectives in flight)
        launch = trainer.sync._launch

        def watched(b):
            launch(b)
            seen.append(_abi.lib.case_get_reserved_cus())

        trainer.sync._launch = watched
        for step in range(nsteps):
            b = {k: v.cuda() for k, v in shard(step, rank).items()}
            losses.append(trainer.train_batch(0, b, "train", opt))
        torch.cuda.synchronize()
        torch.save({"params": {n: p.detach().cpu() for n, p in model.named_parameters()}, "losses": losses,
                    "buckets": len(trainer.sync.buckets), "exposed_ms": trainer.sync.exposed_ms(),
                    "reserved_cus": trainer.sync.reserved_cus, "reserved_during_backward": seen,
                    "reserved_after_step": _abi.lib.case_get_reserved_cus(), "replays": 0 if trainer.graphs is None else trainer.graphs.replays,
                    "ema": {n: t.detach().cpu() for n, t in trainer.ema.shadow.items()}}, os.path.join(out, "rank%d.pt" % rank))
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
