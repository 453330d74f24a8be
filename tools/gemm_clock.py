"""Calibration only: how much of the distance to the 2.5 PFLOP/s bf16 peak is the clock the chip holds under MFMA load.
Runs the 8-wave GEMM (NT forward shape and the TN weight-gradient shape) on zeros, on a constant and on random data for ~2 s each,
sampling the shader clock and the socket power from sysfs while it runs.
    python tools/gemm_clock.py [form prefix, e.g. tn]
"""
import glob
import json
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from case_rg_amd import _abi as A, ops  # noqa: E402


def read(path):
    try:
        with open(path) as f:
            return f.read()
    except OSError:
        return ""


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.sclk = glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk")
        self.power = glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average") + glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input")
        self.on = False
        self.clk, self.pw = [], []
        self.stop = False

    def run(self):
        while not self.stop:
            if self.on:
                clk, pw = None, None
                for p in self.sclk[:1]:
                    for line in read(p).splitlines():
                        if line.rstrip().endswith("*"):
                            clk = float(line.split(":")[1].strip().rstrip("*").strip().lower().replace("mhz", ""))
                for p in self.power[:1]:
                    v = read(p).strip()
                    if v:
                        pw = float(v) / 1e6
                if clk is not None and pw is not None:
                    self.clk.append(clk)
                    self.pw.append(pw)
            time.sleep(0.02)

    def window(self):
        """(mean clock, min clock, mean power) over the samples taken under load: those whose power is within 15 % of the window's maximum
        (the sysfs reads go through the SMU and some of them return only after the queue has drained)."""
        c, p = self.clk, self.pw
        self.clk, self.pw = [], []
        n = min(len(c), len(p))
        if n == 0:
            return None, None, None
        top = max(p[:n])
        busy = [i for i in range(n) if p[i] >= 0.85 * top]
        cb, pb = [c[i] for i in busy], [p[i] for i in busy]
        return round(sum(cb) / len(cb)), round(min(cb)), round(sum(pb) / len(pb))


def main():
    dev, dt = "cuda", torch.bfloat16
    M, K, N = 122880, 2560, 7680
    smp = Sampler()
    smp.start()
    out = []
    for data in ("zeros", "ones", "randn"):
        if data == "randn":
            x = torch.randn(M, K, device=dev).to(dt)
            w = (torch.randn(N, K, device=dev) * K ** -0.5).to(dt)
            y = torch.randn(M, N, device=dev).to(dt)
        else:
            f = torch.zeros if data == "zeros" else torch.ones
            x, w, y = f(M, K, device=dev, dtype=dt), f(N, K, device=dev, dtype=dt), f(M, N, device=dev, dtype=dt)
        dw = torch.zeros(N, K, device=dev)
        o = torch.empty(M, N, device=dev, dtype=dt)
        dwb = torch.empty(N, K, device=dev, dtype=dt)
        forms = {
            "nt M=122880 N=7680 K=2560": lambda: ops.gemm(x, w, o, M, N, K, K, K, N),
            "tn M=7680 N=2560 K=122880 split 5 atomics": lambda: ops.gemm(y, x, dw, N, K, M, N, K, K, a_kmajor=True, b_kmajor=True, split_k=5, epilogue=A.EPI_ATOMIC),
            "vendor nt (torch.matmul, same tensors)": lambda: torch.matmul(x, w.t(), out=o),
            "vendor tn (torch.matmul, bf16 out)": lambda: torch.matmul(y.t(), x, out=dwb),
        }
        for name, fn in forms.items():
            if len(sys.argv) > 1 and not name.startswith(sys.argv[1]):
                continue
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            iters = 800  # ~4 s: the sysfs reads go through the SMU and take tens of milliseconds each
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(iters):
                fn()
                if i == 100:  # the queue is ~1 s deep by now: sample only while it drains
                    torch.cuda.synchronize()
                    smp.window()
                    smp.on = True
            e1.record()
            for _ in range(40):
                time.sleep(0.05)
                if e1.query():
                    break
            smp.on = False
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / iters * 1e-3
            stamp = None
            if hasattr(A.lib, "case_debug_gemm_clock") and not name.startswith("vendor"):  # G8_CLOCK_STAMPS build: exact, from inside the last launch
                import ctypes
                two = (ctypes.c_ulonglong * 2)()
                A.lib.case_debug_gemm_clock.restype = ctypes.c_int
                if A.lib.case_debug_gemm_clock(two) == 0 and two[1]:
                    stamp = round(two[0] / two[1] * 100.0)
            clk, clk_min, pw = smp.window()
            rec = {"data": data, "form": name, "ms": round(t * 1e3, 3), "tflops": round(2.0 * M * N * K / t / 1e12, 1), "sclk_mhz_avg": clk, "sclk_mhz_min": clk_min, "power_w_avg": pw}
            if stamp:
                rec["sclk_mhz_in_kernel"] = clk = stamp  # s_memtime / s_memrealtime (100 MHz) inside workgroup 0 of the last launch
            if clk:
                rec["peak_at_that_clock_tflops"] = round(256 * 4 * 1024 * clk * 1e6 / 1e12, 1)
                rec["frac_of_peak_at_that_clock"] = round(rec["tflops"] / rec["peak_at_that_clock_tflops"], 3)
            print(json.dumps(rec), flush=True)
            out.append(rec)
    smp.stop = True
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/gemm_clock.json", "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
