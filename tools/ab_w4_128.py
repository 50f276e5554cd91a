"""A/B of the four-wave 128x128 fused tile (variant 20) against the 8-wave one (9) and the automatic choice on single-tenant mid-M shapes."""
import sys
sys.path.insert(0, ".")
import torch
import bitdelta_amd as bd
from bitdelta_amd import _lib
L = _lib.lib()
g = torch.Generator(device="cuda").manual_seed(0)
big = len(sys.argv) > 1 and sys.argv[1] == "big"
for T, M in (((1, 1024), (1, 1536), (1, 2048)) if big else ((1, 192), (1, 256), (1, 512), (1, 768), (2, 128), (6, 128), (6, 96))):
    for name, N, K in (("o", 4096, 4096), ("qkv-mistral", 6144, 4096), ("qkv", 12288, 4096), ("gate+up", 22016, 4096), ("down", 4096, 11008)):
        x = torch.randn(T, M, K, device="cuda", generator=g).bfloat16()
        w = (torch.randn(N, K, device="cuda", generator=g) * 0.02).bfloat16()
        mask = torch.randint(-2**31, 2**31 - 1, (T, K // 32, N), device="cuda", generator=g, dtype=torch.int64).to(torch.int32)
        alpha = torch.full((T, 1), 4e-4, device="cuda")
        row = []
        for v in (-1, 9, 20, 14):
            L.bd_set_gemm_variant(v)
            try:
                for _ in range(3):
                    bd.binary_linear(x, w, mask, alpha)
                used = L.bd_last_gemm_variant()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    bd.binary_linear(x, w, mask, alpha)
                e1.record(); torch.cuda.synchronize()
                us = e0.elapsed_time(e1) / 20 * 1e3
                row.append(f"v{v}->{used}: {us:7.1f} us")
            except Exception as e:
                row.append(f"v{v}: {type(e).__name__}")
            finally:
                L.bd_set_gemm_variant(-1)
        print(f"T={T} M={M:4d} {name:11s} | " + " | ".join(row), flush=True)
