#!/usr/bin/env python3
"""RMSNorm launches at prefill sizes: the wave-per-row kernel (bd_srv_rmsnorm from 64 rows on) against stock torch (F.rms_norm with the weight
fused = what bench_model used; F.rms_norm + weight multiply = what the serving loop used beyond 1024 rows).  hipGraph of 20 calls, median."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from bitdelta_amd import serving_ops as ops


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(20):
                fn()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s); g.replay(); e1.record(s); s.synchronize()
            ts.append(e0.elapsed_time(e1) / 20 * 1e3)
    return sorted(ts)[len(ts) // 2]


for T, M, H in ((1, 2048, 4096), (1, 1024, 4096), (6, 64, 4096), (6, 256, 4096), (1, 2048, 8192)):
    xs = [torch.randn(T, M, H, device="cuda").bfloat16() for _ in range(8)]
    w = (1 + 0.1 * torch.randn(T, H, device="cuda")).bfloat16()
    it = [0]
    def nx():
        it[0] += 1
        return xs[it[0] % 8]
    a = timed(lambda: ops.rmsnorm_tenant(nx(), w, 1e-5))
    b = timed(lambda: F.rms_norm(nx(), (H,), None, 1e-5) * w[:, None, :])
    c = timed(lambda: F.rms_norm(nx().view(1, -1, H), (H,), w[0], 1e-5)) if T == 1 else float("nan")
    print(f"T={T} M={M:5d} H={H}: HIP wave-per-row {a:6.2f} us | torch rms_norm * w {b:6.2f} us | torch rms_norm(weight) {c:6.2f} us   ({2 * T * M * H * 2 / a / 1e6:5.2f} TB/s)")
