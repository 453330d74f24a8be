"""Feasibility probe (round 6): capture the WHOLE training step (forward + backward + clip + Adam + EMA) of bench.py's model into one
hipGraph with the library as it is and time eager against replay.  The replay redraws the same dropout masks and keeps the Adam
scalars of the captured step -- it measures what a captured step would cost, it is not a training loop.

    python tools/graph_probe.py --mode refdefault
    python tools/graph_probe.py --model masque --batch 8
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    a = bench.parse()
    if a.mode == "refdefault":
        a.hidden, a.passages, a.passage_len, a.query_len, a.answer_len, a.enc_layers = 256, 10, 100, 60, 40, 3
        a.batch = 16 if a.batch == 32 else a.batch
        a.mode = "train"
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    trainer, opt, sched, batch = bench.build(a, device)
    host = torch.empty(3 if a.model == "case" else 2, dtype=torch.float32).pin_memory()

    def body():
        loss = trainer.model(dict(batch), method="train")
        parts = torch.cat([l.mean().reshape(1) for l in loss])
        parts.sum().backward()
        host[:parts.numel()].copy_(parts.detach().float(), non_blocking=True)
        opt.step(clip_norm=1.0, ema=trainer.ema)
        opt.zero_grad()

    def timed(fn, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
            torch.cuda.synchronize()  # the trainer hands the losses to the host every step
        return (time.perf_counter() - t0) / n * 1e3

    for _ in range(3):
        body()
    eager = timed(body, a.steps)
    eager_losses = host.tolist()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            body()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    t0 = time.perf_counter()
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        body()
    capture_s = time.perf_counter() - t0
    opt._stage = [(s, None) for s, _ in opt._stage]  # events recorded inside the capture cannot be synchronised
    replay = timed(graph.replay, a.steps)
    print(json.dumps({"probe": "whole-step hipGraph", "model": a.model, "hidden": a.hidden, "batch": a.batch, "eager_ms": round(eager, 3),
                      "replay_ms": round(replay, 3), "capture_s": round(capture_s, 2), "eager_losses": eager_losses, "replay_losses": host.tolist(),
                      "pool_gb": round(torch.cuda.memory_reserved() / 2 ** 30, 2)}))


if __name__ == "__main__":
    main()
