#!/usr/bin/env python3
"""Calibration of the "power-limited ceiling" claim: sustained runs of the vendor's dense bf16 GEMM (torch.matmul ->
hipBLASLt) and of the W1A16 delta-GEMM (bd_delta_bmm -> delta_gemm_w4_kernel) at the SAME shapes, K = N = 4096, M in {4096, 8192,
16384}, in ONE process on ONE board, alternating, with board power and shader clock sampled from hwmon at 10 Hz.

    python tools/vendor_gemm.py [seconds per run, default 2.5]  > profiles/r04_vendor_gemm.txt

Both kernels execute 2*M*N*K MFMA flops; the delta-GEMM's second operand is 1/16 of the bytes.  If the vendor GEMM reaches >= 0.70
of 2.5 PF the ceiling argument is void; if it sits at or below the delta-GEMM at the same ~1.3 kW, 2.5 PF is not reachable by a dense
bf16 GEMM at this board's power cap and the roofline fraction should be read against what the board can sustain.
"""
import glob
import sys
import threading
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bitdelta_amd as bd  # noqa: E402

PEAK = 2500.0


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.hw = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
        self.rows = []
        self.stop = False

    @staticmethod
    def _read(path):
        try:
            with open(path) as fh:
                return float(fh.read().strip())
        except (OSError, ValueError):
            return float("nan")

    def run(self):
        while not self.stop:
            t = time.perf_counter()
            for h in self.hw:
                p = self._read(h + "/power1_input")
                if p != p:
                    p = self._read(h + "/power1_average")
                self.rows.append((t, h, p * 1e-6, self._read(h + "/freq1_input") * 1e-6))
            time.sleep(0.1)

    def window(self, t0, t1):
        """(hwmon, avg W, max W, avg MHz, samples) of the card whose power is highest inside [t0, t1]"""
        best = None
        for h in self.hw:
            r = [(p, f) for (t, hh, p, f) in self.rows if hh == h and t0 <= t <= t1 and p == p]
            if not r:
                continue
            avg = sum(p for p, _ in r) / len(r)
            if best is None or avg > best[1]:
                best = (h, avg, max(p for p, _ in r), sum(f for _, f in r) / len(r), len(r))
        return best


def sustained(fn, seconds):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        n += 50
    return n, t0, time.perf_counter()


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 2.5
    dev = torch.device("cuda:0")
    smp = Sampler()
    smp.start()
    time.sleep(1.0)
    idle = smp.window(0, time.perf_counter())
    print(f"# idle: {idle}")
    N = K = 4096
    print("# kernel               M      us/launch   TFLOP/s   frac of 2.5 PF   avg W   max W   avg sclk MHz   samples")
    for rep in range(2):
        for M in (4096, 8192, 16384):
            g = torch.Generator(device=dev).manual_seed(1)
            x = torch.randn(1, M, K, device=dev, generator=g).bfloat16()
            w = (torch.randn(N, K, device=dev, generator=g) * 0.02).bfloat16()
            p = torch.randint(-2 ** 31, 2 ** 31 - 1, (1, K // 32, N), device=dev, generator=g, dtype=torch.int64).to(torch.int32)
            out = torch.empty(1, M, N, device=dev, dtype=torch.bfloat16)
            x2, wt, out2 = x[0], w.t(), out[0]
            for name, fn in (("vendor torch.matmul", lambda: torch.matmul(x2, wt, out=out2)),
                             ("delta_gemm (w4)    ", lambda: bd.delta_bmm(x, p, out=out, round_mode=0))):
                n, t0, t1 = sustained(fn, secs)
                us = (t1 - t0) / n * 1e6
                tf = 2.0 * M * N * K / us * 1e-6
                wv = smp.window(t0 + 0.3, t1)
                print(f"{name}  {M:6d}   {us:9.2f}   {tf:7.1f}   {tf / PEAK:8.3f}        "
                      f"{wv[1]:6.0f}  {wv[2]:6.0f}   {wv[3]:8.0f}      {wv[4]}" if wv else f"{name} {M} {us:.2f} us {tf:.1f} TF (no hwmon)")
                time.sleep(0.5)
            del x, w, p, out
    smp.stop = True


if __name__ == "__main__":
    main()
