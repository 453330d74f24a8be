"""Probe for a two-lane greedy step: does a half-batch K21 launch (128 items, ONE workgroup per item = half of the CUs) overlap with the other
lane's chain of small row-local launches (M = 128 projections + LayerNorms) when the two run on different streams?
Prints: K21 alone, the chain alone, both on one stream, both on two streams (ms per iteration of 4 x K21 + 48 projections + 26 LayerNorms)."""
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from case_rg_amd import _abi as A  # noqa: E402
from case_rg_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dt = torch.bfloat16
S, E = 3840, 512


def make(B):
    mem = torch.randn(B, S, E, device=dev).to(dt)
    qp = torch.randn(B, 8 * E, device=dev).to(dt)
    out = torch.empty(B, 8 * E, device=dev, dtype=dt)
    x = torch.randn(B, E, device=dev).to(dt)
    w = (torch.randn(E, E, device=dev) * E ** -0.5).to(dt)
    b = torch.zeros(E, device=dev)
    g = torch.ones(E, device=dev)
    return dict(B=B, mem=mem, qp=qp, out=out, x=x, w=w, b=b, g=g)


def k21(d, nsplit):
    B = d["B"]
    need = A.lib.case_attention_decode_mqa_workspace(B, S, nsplit)
    if need and "ws" not in d:
        d["ws"] = torch.empty(need // 4, dtype=torch.float32, device=dev)
    ws = d.get("ws")
    A.call("case_attention_decode_mqa", ops._ptr(d["qp"]), ops._ptr(d["mem"]), None, ops._ptr(d["out"]), B, S, 8 * E, nsplit, ops._ptr(ws), need, ops._stream())


def chain(d, n_gemm, n_ln):
    x = d["x"]
    for i in range(n_gemm):
        x = ops.linear(x, d["w"], d["b"])
        if i < n_ln:
            x = ops.layer_norm(x, d["g"], d["b"], 1e-5)
    return x


def timed(fn, rep=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rep):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / rep


full = make(256)
print("B 256: 4 x K21 %.3f ms   chain (48 projections + 26 LayerNorms) %.3f ms   both, one stream %.3f ms"
      % (timed(lambda: [k21(full, 1) for _ in range(4)]), timed(lambda: chain(full, 48, 26)),
         timed(lambda: ([k21(full, 1) for _ in range(4)], chain(full, 48, 26)))), flush=True)
la, lb = make(128), make(128)
for nsplit in (1, 2):
    t_k = timed(lambda: [k21(la, nsplit) for _ in range(4)])
    t_c = timed(lambda: chain(lb, 48, 26))
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def two():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur)
        s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            for _ in range(4):
                k21(la, nsplit)
        with torch.cuda.stream(s2):
            chain(lb, 48, 26)
        cur.wait_stream(s1)
        cur.wait_stream(s2)

    def lanes():  # what a two-lane step does: each lane runs its K21s AND its chain, lane by lane on its own stream
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur)
        s2.wait_stream(cur)
        for s, d in ((s1, la), (s2, lb)):
            with torch.cuda.stream(s):
                for _ in range(4):
                    chain(d, 12, 6)
                    k21(d, nsplit)
        cur.wait_stream(s1)
        cur.wait_stream(s2)

    g = torch.cuda.CUDAGraph()
    lanes()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        lanes()
    print("B 128 per lane, K21 splits %d: 4 x K21 alone %.3f ms   chain alone %.3f ms   K21 || chain on two streams %.3f ms   "
          "two full lanes (each 4 x [12 proj + 6 LN + K21]) eager %.3f ms, graph %.3f ms"
          % (nsplit, t_k, t_c, timed(two), timed(lanes), timed(g.replay)), flush=True)


def one_lane_full():
    for _ in range(4):
        chain(full, 12, 6)
        k21(full, 1)


g1 = torch.cuda.CUDAGraph()
one_lane_full()
torch.cuda.synchronize()
with torch.cuda.graph(g1):
    one_lane_full()
print("B 256 one lane (4 x [12 proj + 6 LN + K21]): eager %.3f ms, graph %.3f ms" % (timed(one_lane_full), timed(g1.replay)), flush=True)
