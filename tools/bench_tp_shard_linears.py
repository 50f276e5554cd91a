#!/usr/bin/env python3
"""The four fused Linears of ONE rank's Llama-2-70B TP = 8 shard at prefill (M = 2048): automatic dispatch against forced tile variants
(14 = four-wave 256x128, 20 = four-wave 128x128, 8 / 9 = the 8-wave forms, 10 = 128x128 split-k + reduce).  python tools/bench_tp_shard_linears.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bitdelta_amd as bd
from bitdelta_amd import _lib

L = _lib.lib()
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
shapes = [("q|k|v shard", 1280, 8192), ("o shard", 8192, 1024), ("gate|up shard", 7168, 8192), ("down shard", 8192, 3584)]
for name, N, K in shapes:
    x = torch.randn(1, M, K, device=dev, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev, generator=g) * 0.02).to(torch.bfloat16)
    mask = torch.randint(-2**31, 2**31 - 1, (1, K // 32, N), device=dev, generator=g, dtype=torch.int64).to(torch.int32)
    alpha = torch.full((1, 1), 4e-4, device=dev)
    row = []
    for v in (-1, 14, 20, 8, 9, 10):
        L.bd_set_gemm_variant(v)
        try:
            for _ in range(3):
                y = bd.binary_linear(x, w, mask, alpha)
            used = L.bd_last_gemm_variant()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                y = bd.binary_linear(x, w, mask, alpha)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 20 * 1e3
            row.append(f"v{v}->{used}: {us:7.1f} us {4.0 * M * N * K / us * 1e-6:5.0f} TF")
        except Exception as e:
            row.append(f"v{v}: {type(e).__name__}")
        finally:
            L.bd_set_gemm_variant(-1)
    print(f"M={M} {name:14s} N={N:5d} K={K:5d} | " + " | ".join(row), flush=True)
