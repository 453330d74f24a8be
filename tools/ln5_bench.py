"""LayerNorm<5H> backward + concat5 backward: the two kernels against the fused one (case_layernorm_bwd_concat5) at cfg 2's shape."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from case_rg_amd import _abi as A

def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

rows, H = 122880, 512
C = 5 * H
bf = torch.bfloat16
x = torch.randn(rows, C, device="cuda").to(bf)
dy = torch.randn(rows, C, device="cuda").to(bf)
add = torch.randn(rows, C, device="cuda").to(bf)
gamma = torch.randn(C, device="cuda")
mean = x.float().mean(1).contiguous(); rstd = (x.float().var(1, unbiased=False) + 1e-5).rsqrt().contiguous()
valid = torch.ones(rows, dtype=torch.uint8, device="cuda")
dx = torch.empty_like(x)
de, d1, d2 = (torch.empty(rows, H, device="cuda", dtype=bf) for _ in range(3))
dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
e, a1, a2 = x[:, :H].contiguous(), x[:, H:2 * H].contiguous(), x[:, 2 * H:3 * H].contiguous()

def two(addp):
    A.call("case_layernorm_bwd", dy.data_ptr(), x.data_ptr(), None, gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(), addp, dg.data_ptr(), db.data_ptr(), rows, C, A.BF16, 0)
    A.call("case_concat5_bwd", dx.data_ptr(), e.data_ptr(), a1.data_ptr(), a2.data_ptr(), valid.data_ptr(), de.data_ptr(), d1.data_ptr(), d2.data_ptr(), rows, H, A.BF16, 0)

def one(addp):
    A.call("case_layernorm_bwd_concat5", dy.data_ptr(), x.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(), addp, x.data_ptr(), x.data_ptr(), x.data_ptr(),
           valid.data_ptr(), de.data_ptr(), d1.data_ptr(), d2.data_ptr(), dg.data_ptr(), db.data_ptr(), rows, H, A.BF16, 0)

for name, addp in (("with dx_add", add.data_ptr()), ("without", None)):
    t2, t1 = timeit(lambda: two(addp)), timeit(lambda: one(addp))
    gb = (rows * C * 2 * (3 if addp else 2) + 3 * rows * H * 2) / 1e9
    print("%s: two kernels %.3f ms, fused %.3f ms (%.2f TB/s on %.2f GB)" % (name, t2, t1, gb / t1, gb), flush=True)

# the dual-output backward (dx + dropout-masked copy) at 5H: 2.5 GB per call
g2 = torch.empty_like(x)
def dual():
    A.call("case_layernorm_bwd_dropout", dy.data_ptr(), x.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(), g2.data_ptr(),
           dg.data_ptr(), db.data_ptr(), rows, C, 0.1, 1234, 0, None, A.BF16, 0)
def plain_then_drop():
    A.call("case_layernorm_bwd", dy.data_ptr(), x.data_ptr(), None, gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(), None, dg.data_ptr(), db.data_ptr(), rows, C, A.BF16, 0)
    A.call("case_dropout", dx.data_ptr(), g2.data_ptr(), x.numel(), 0.1, 1234, 0, None, A.BF16, 0)
t1, t2 = timeit(dual), timeit(plain_then_drop)
print("5H dual-output LayerNorm backward %.3f ms (%.2f TB/s on 2.52 GB); LayerNorm backward + dropout pass %.3f ms" % (t1, 2.52 / t1, t2), flush=True)
