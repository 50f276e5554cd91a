#!/usr/bin/env python3
"""Per-launch A/B of the streaming decode kernel's options on the Mistral-7B decode shapes, cold weights (the copies rotate through > 2x the
Infinity Cache), through the shipped library: bd_set_stream_tuning flags 16 / 32 = nt weight loads on / off, 64 / 128 = resident activation
rows on / off.  (Round 4 also tried NS 6 for the plain 6-tenant form: 5-14 % slower, and NS 8 for the resident form: equal.)
Small launches are host-bound here (Python call ~17 us): read the large shapes; tests/native/ring_bench has the per-launch table.
usage: python tools/ab_decode_depth.py [T]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bitdelta_amd import _lib
from bitdelta_amd.binary_gemm_kernel import binary_linear_decode, pack_decode_masks, tile_weight

T = int(sys.argv[1]) if len(sys.argv) > 1 else 6
L = _lib.lib()
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
shapes = [("o", 4096, 4096, False), ("q|k|v", 6144, 4096, False), ("gate|up->swiglu", 28672, 4096, True), ("down", 4096, 14336, False)]
cfgs = [("default", 32 | 128), ("nt", 16 | 128), ("nt+xres", 16 | 64), ("xres", 32 | 64)]
for name, N, K, sw in shapes:
    nbytes = 2.0 * N * K + T * N * K / 8
    nset = max(2, int(600e6 / nbytes) + 1)
    ws, ms = [], []
    for _ in range(nset):
        w = (torch.randn(N, K, device=dev, generator=g) * 0.02).half()
        ws.append(tile_weight(w))
        ms.append(pack_decode_masks(torch.randint(-2**31, 2**31 - 1, (T, K // 32, N), device=dev, generator=g, dtype=torch.int64).to(torch.int32)))
        del w
    x = torch.randn(T, 1, K, device=dev, generator=g).half()
    alpha = torch.full((T, 2 if sw else 1), 4e-4, device=dev)
    row = []
    for cname, flag in cfgs:
        L.bd_set_stream_tuning(flag)
        try:
            def call(i):
                return binary_linear_decode(x, ws[i % nset], ms[i % nset], alpha, layout="packed", groups=2 if sw else 1, swiglu=sw, weight_tiled=True)
            for i in range(nset + 3):
                call(i)
            best = 1e9
            for rep in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(40):
                    call(i)
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 40 * 1e3)
            row.append(f"{cname}: {best:6.2f} us {nbytes / best * 1e-3:5.0f} GB/s")
        finally:
            L.bd_set_stream_tuning(0)
    print(f"T={T} {name:16s} {nbytes * 1e-6:6.1f} MB | " + " | ".join(row), flush=True)
    del ws, ms
