#!/usr/bin/env python3
"""Multi-tenant prefill of short prompts (demo_backend.py:297-299: prompts left-padded to 64 .. 1024 tokens, T tenants): the fused
Linear with per-tenant masks at M = 64 .. 256 on the Mistral-7B shapes, automatic dispatch vs forced tile variants.
usage: python tools/bench_mt_prefill.py [T]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bitdelta_amd as bd
from bitdelta_amd import _lib

T = int(sys.argv[1]) if len(sys.argv) > 1 else 6
L = _lib.lib()
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
shapes = [("q+k+v", 6144, 4096), ("o", 4096, 4096), ("gate+up", 28672, 4096), ("down", 4096, 14336)]
for M in (32, 64, 128) if len(sys.argv) <= 2 else [int(a) for a in sys.argv[2:]]:
    for name, N, K in shapes:
        x = torch.randn(T, M, K, device=dev, generator=g).to(torch.float16)
        w = (torch.randn(N, K, device=dev, generator=g) * 0.02).to(torch.float16)
        mask = torch.randint(-2**31, 2**31 - 1, (T, K // 32, N), device=dev, generator=g, dtype=torch.int64).to(torch.int32)
        alpha = torch.full((T, 1), 4e-4, device=dev)
        row = []
        for v in (-1, 16, 17, 18, 19, 20, 9):
            if v in (11, 12, 16, 17, 18, 19) and M > 64:
                continue
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
                row.append(f"v{v}->{used}: {us:7.1f} us {4.0 * T * M * N * K / us * 1e-6:6.0f} TF")
            except Exception as e:
                row.append(f"v{v}: {type(e).__name__}")
            finally:
                L.bd_set_gemm_variant(-1)
        print(f"T={T} M={M:4d} {name:8s} N={N:5d} K={K:5d} | " + " | ".join(row), flush=True)
