"""Micro-benchmark of the LayerNorm / column-sum kernels at the cfg 2 shapes (achieved HBM GB/s on algorithmic bytes)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from case_rg_amd import _abi as A, ops  # noqa: E402


def timeit(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    dt = torch.bfloat16
    for rows, cols in ((122880, 512), (122880, 2560), (2048, 512)):
        x = torch.randn(rows, cols, device="cuda").to(dt)
        dy = torch.randn(rows, cols, device="cuda").to(dt)
        extra = torch.randn(rows, cols, device="cuda").to(dt)
        g, b = torch.randn(cols, device="cuda"), torch.randn(cols, device="cuda")
        y, dx = torch.empty_like(x), torch.empty_like(x)
        mean, rstd = torch.empty(rows, device="cuda"), torch.empty(rows, device="cuda")
        dg, db = torch.zeros(cols, device="cuda"), torch.zeros(cols, device="cuda")
        s = torch.cuda.current_stream().cuda_stream
        p = lambda t: None if t is None else t.data_ptr()
        nb = rows * cols * 2
        t = timeit(lambda: A.call("case_layernorm_fwd", p(x), None, p(g), p(b), p(y), p(mean), p(rstd), rows, cols, 1e-5, A.BF16, s))
        print(json.dumps({"kernel": "ln_fwd", "rows": rows, "cols": cols, "us": round(t * 1e6, 1), "GBps": round(2 * nb / t / 1e9)}))
        for add in (None, extra):
            t = timeit(lambda: A.call("case_layernorm_bwd", p(dy), p(x), None, p(g), p(mean), p(rstd), p(dx), p(add), p(dg), p(db), rows, cols, A.BF16, s))
            nbytes = (4 if add is not None else 3) * nb
            print(json.dumps({"kernel": "ln_bwd" + ("+add" if add is not None else ""), "rows": rows, "cols": cols, "us": round(t * 1e6, 1), "GBps": round(nbytes / t / 1e9)}))
        out = torch.zeros(cols, device="cuda")
        t = timeit(lambda: A.call("case_colsum", p(dy), p(out), rows, cols, A.BF16, s))
        print(json.dumps({"kernel": "colsum", "rows": rows, "cols": cols, "us": round(t * 1e6, 1), "GBps": round(nb / t / 1e9)}))
        t = timeit(lambda: A.call("case_add", p(x), p(dy), p(y), rows * cols, A.BF16, s))
        print(json.dumps({"kernel": "add (3 streams, reference)", "rows": rows, "cols": cols, "us": round(t * 1e6, 1), "GBps": round(3 * nb / t / 1e9)}))


if __name__ == "__main__":
    main()
