#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r4h; mkdir -p $O
timeout 600 python tools/bench_mt_prefill.py 6 64 32 > $O/mt_prefill.txt 2>&1; echo rc=$?; cat $O/mt_prefill.txt
timeout 600 python - > $O/pair_check.txt 2>&1 <<'P'
# correctness of the pair variants vs the per-tenant 64-row tiles (variant 11) and the oracle on sampled columns
import torch, sys
sys.path.insert(0, '.')
import bitdelta_amd as bd
from bitdelta_amd import _lib
from oracle import bd_oracle as o
L = _lib.lib()
torch.manual_seed(0)
for (T, M, N, K) in [(6, 64, 4096, 4096), (6, 37, 1024, 4096), (5, 64, 6144, 4096), (2, 17, 256, 512), (6, 64, 4096, 14336), (3, 64, 28672, 4096)]:
    for dt in (torch.float16, torch.bfloat16):
        x = torch.randn(T, M, K).to(dt); w = (torch.randn(N, K) * 0.02).to(dt)
        p = torch.randint(-2**31, 2**31 - 1, (T, K // 32, N), dtype=torch.int64).to(torch.int32)
        al = (torch.rand(T, 1) * 2e-4 + 3e-4)
        cols = torch.randint(0, N, (24,)).unique()
        ref = o.binary_linear(x, w[cols].contiguous(), p[:, :, cols].contiguous(), al, out_dtype=torch.float32)
        xd, wd, pd, ad = x.cuda(), w.cuda(), p.cuda(), al.cuda()
        L.bd_set_gemm_variant(11); y11 = bd.binary_linear(xd, wd, pd, ad); L.bd_set_gemm_variant(-1)
        for v in (16, 17):
            L.bd_set_gemm_variant(v)
            try:
                y = bd.binary_linear(xd, wd, pd, ad); used = L.bd_last_gemm_variant()
                y32 = bd.binary_linear(xd, wd, pd, ad, out_dtype=torch.float32) if v == 16 else None
            finally:
                L.bd_set_gemm_variant(-1)
            got = y[:, :, cols.cuda()].float().cpu()
            rel = ((got - ref).norm() / ref.norm()).item()
            same = torch.equal(y, y11)
            r32 = ((y32[:, :, cols.cuda()].cpu().double() - ref.double()).norm() / ref.double().norm()).item() if y32 is not None else -1
            print(f"T={T} M={M} N={N} K={K} {str(dt)[6:]} v{v}->{used}: rel vs oracle {rel:.2e}  fp32-mode {r32:.2e}  bit-identical to v11: {same}")
P
cat $O/pair_check.txt
