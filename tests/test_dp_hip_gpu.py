"""Data parallelism with the HIP model across two ranks: two fresh child processes on device 0, gloo
collectives, the real trainer loop (CumulativeTrainer.train_batch -> GradSync hooks / buckets -> FusedAdam reading the bucket-view
gradients -> EMA -> zero_grad).  After three steps the parameters (and EMA shadows) are bit-identical across the ranks and equal a
one-process run on the concatenated batches within the fp32 bar; the ranks started from different weights, so the broadcast at
construction is under test too.  (reference: common/CumulativeTrainer.py:45-47,52-78)"""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.timeout(600)
def test_two_ranks_of_the_hip_model_stay_bit_identical_and_match_one_process(tmp_path):
    import dp_hip_worker as W
    port = str(_free_port())
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_hip_worker.py"), str(r), "2", port, str(tmp_path)], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=500)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-2000:]
    r0, r1 = torch.load(tmp_path / "rank0.pt"), torch.load(tmp_path / "rank1.pt")
    assert r0["buckets"] == r1["buckets"] >= 1 and r0["reserved_cus"] == 8, "GradSync reserves one CU per XCD for the collectives by default"
    # ... and holds the reservation only while collectives are in flight: seen by a gradient hook late in backward, released by finish()
    assert r0["reserved_during_backward"] and all(v == 8 for v in r0["reserved_during_backward"]) and r0["reserved_after_step"] == 0
    for n in r0["params"]:
        assert torch.equal(r0["params"][n], r1["params"][n]), "ranks drifted apart on " + n
        assert torch.equal(r0["ema"][n], r1["ema"][n]), "EMA shadows drifted apart on " + n
    # one process, the concatenated batches (equal shard sizes and full-length sequences: the mean of the shard means IS the mean)
    from case_rg_amd.common.CumulativeTrainer import CumulativeTrainer
    from case_rg_amd.optim import FusedAdam
    model = W.build(40)
    trainer = CumulativeTrainer(model, None, None, 0, 1)
    assert trainer.sync is None
    opt = FusedAdam(model.parameters(), lr=1e-3)
    losses = []
    for step in range(3):
        a, b = W.shard(step, 0), W.shard(step, 1)
        both = {k: torch.cat([a[k], b[k]], dim=0).cuda() for k in a}
        losses.append(trainer.train_batch(0, both, "train", opt))
    torch.cuda.synchronize()
    worst = 0.0
    for n, p in model.named_parameters():
        want, got = p.detach().cpu(), r0["params"][n]
        err = (got - want).abs().max().item() / (want.abs().max().item() + 1e-12)
        worst = max(worst, err)
        assert err <= 3e-3, "%s: %.2e" % (n, err)   # three Adam steps of lr 1e-3: the update g / sqrt(v) turns the last-ulp differences of f32 sums in another order into ~1e-4 of a step (measured 7.5e-4 of the tensor scale)
    for step in range(3):
        mean = [(x + y) / 2 for x, y in zip(r0["losses"][step], r1["losses"][step])]
        assert all(abs(m - w) <= 1e-4 * max(1.0, abs(w)) for m, w in zip(mean, losses[step])), (step, mean, losses[step])
    print("two-rank HIP DP: worst parameter deviation from the one-process run %.2e" % worst)


@pytest.mark.timeout(600)
def test_two_ranks_with_captured_steps_stay_bit_identical_and_match_one_process(tmp_path):
    """The same two-rank loop with hipGraph-captured steps (round 6): forward + backward and clip + Adam + EMA replay from two graphs,
    GradSync's bucketed all-reduce runs eagerly between them on the gradients the first graph left behind.  Six steps (two eager, one
    recorded + replayed, three more replays): ranks bit-identical, equal to the one-process EAGER run on the concatenated batches."""
    import dp_hip_worker as W
    port = str(_free_port())
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_hip_worker.py"), str(r), "2", port, str(tmp_path), "graph"], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=500)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    r0, r1 = torch.load(tmp_path / "rank0.pt"), torch.load(tmp_path / "rank1.pt")
    assert r0["replays"] == r1["replays"] == 4
    for n in r0["params"]:
        assert torch.equal(r0["params"][n], r1["params"][n]), "ranks drifted apart on " + n
        assert torch.equal(r0["ema"][n], r1["ema"][n]), "EMA shadows drifted apart on " + n
    from case_rg_amd.common.CumulativeTrainer import CumulativeTrainer
    from case_rg_amd.optim import FusedAdam
    model = W.build(40)
    trainer = CumulativeTrainer(model, None, None, 0, 1)
    opt = FusedAdam(model.parameters(), lr=1e-3)
    losses = []
    for step in range(6):
        a, b = W.shard(step, 0), W.shard(step, 1)
        losses.append(trainer.train_batch(0, {k: torch.cat([a[k], b[k]], dim=0).cuda() for k in a}, "train", opt))
    torch.cuda.synchronize()
    worst = 0.0
    for n, p in model.named_parameters():
        want, got = p.detach().cpu(), r0["params"][n]
        err = (got - want).abs().max().item() / (want.abs().max().item() + 1e-12)
        worst = max(worst, err)
        assert err <= 6e-3, "%s: %.2e" % (n, err)  # (six Adam steps; the three-step eager run above measures 7.5e-4)
    for step in range(6):
        mean = [(x + y) / 2 for x, y in zip(r0["losses"][step], r1["losses"][step])]
        assert all(abs(m - w) <= 2e-4 * max(1.0, abs(w)) for m, w in zip(mean, losses[step])), (step, mean, losses[step])
    print("two-rank HIP DP with captured steps: worst parameter deviation from the one-process run %.2e" % worst)
