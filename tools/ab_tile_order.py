#!/usr/bin/env python3
"""Tile walk order of the fused prefill GEMM (GemmParams::group_m via bd_set_tile_group_m): time per launch at the four fused Llama-2-7B
shapes of the timed step (M = 2048, bf16), group_m = 1 (n fastest) / 2 / 4 (shipped for fused launches) / 8 (m fastest at 8 tile rows).
With `--one GM SHAPE` runs only that configuration 30 times (for a `rocprofv3 --pmc FETCH_SIZE` pass around the process)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bitdelta_amd import _lib
from bitdelta_amd.binary_gemm_kernel import binary_linear

L = _lib.lib()
dev = "cuda"
M, hid, inter = 2048, 4096, 11008
shapes = {"qkv": (3 * hid, hid, 3), "o": (hid, hid, 1), "gateup": (2 * inter, hid, 2), "down": (hid, inter, 1)}
g = torch.Generator(device=dev).manual_seed(0)


def problem(name):
    N, K, G = shapes[name]
    x = torch.randn(1, M, K, device=dev, generator=g).bfloat16()
    w = (torch.randn(N, K, device=dev, generator=g) * 0.02).bfloat16()
    p = torch.randint(-2**31, 2**31 - 1, (1, K // 32, N), device=dev, generator=g, dtype=torch.int64).to(torch.int32)
    a = torch.full((1, G), 4e-4, device=dev)
    return x, w, p, a, G


def run(name, gm, iters):
    x, w, p, a, G = problem(name)
    L.bd_set_tile_group_m(gm)
    try:
        for _ in range(5):
            binary_linear(x, w, p, a, groups=G)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            binary_linear(x, w, p, a, groups=G)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3, L.bd_last_gemm_variant()
    finally:
        L.bd_set_tile_group_m(0)


if len(sys.argv) > 3 and sys.argv[1] == "--one":
    us, v = run(sys.argv[3], int(sys.argv[2]), 30)
    print(f"{sys.argv[3]} group_m {sys.argv[2]}: {us:.1f} us (variant {v})")
else:
    for name in shapes:
        N, K, _ = shapes[name]
        row = []
        for gm in (1, 2, 4, 8):
            us, v = run(name, gm, 40)
            row.append(f"group_m {gm}: {us:7.1f} us {4.0 * M * N * K / us * 1e-6:6.0f} TF")
        print(f"{name:7s} N={N:6d} K={K:6d} variant {v} | " + " | ".join(row), flush=True)
