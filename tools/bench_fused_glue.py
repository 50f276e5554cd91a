#!/usr/bin/env python3
"""Same-process A/B of the decode launches with and without the fused RMSNorm prologue / SwiGLU epilogue, cold weights (the weight
sets are rotated over > 600 MB so neither L2 nor the Infinity Cache helps).  usage: python tools/bench_fused_glue.py [T]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bitdelta_amd import serving_ops as ops
from bitdelta_amd.serving_loop import FusedDeltaLinear

T = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev, dt = "cuda", torch.float16
g = torch.Generator(device=dev).manual_seed(0)
K = 4096


def make(widths, interleave8=False):
    ws = [(torch.randn(n, K, device=dev, generator=g) * 0.02).to(dt) for n in widths]
    ms = [torch.randint(-2**31, 2**31 - 1, (T, K // 32, n), device=dev, generator=g, dtype=torch.int64).to(torch.int32) for n in widths]
    cs = [torch.rand(T, device=dev, generator=g) * 1e-3 for _ in widths]
    return FusedDeltaLinear(ws, ms, cs, interleave8=interleave8)


def timeit(fn, sets, reps=30):
    for i in range(len(sets)):
        fn(sets[i])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(sets[i % len(sets)])
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


x = (torch.randn(T, 1, K, device=dev, generator=g)).to(dt)
nw = (1 + 0.1 * torch.randn(T, K, device=dev, generator=g)).to(dt)
for name, widths, il, nset in (("q+k+v", [4096, 1024, 1024], False, 9), ("gate+up", [14336, 14336], True, 2)):
    sets = [make(widths, il) for _ in range(nset)]
    mb = sets[0].linear_bytes() / 1e6
    plain = timeit(lambda m: m(x), sets)
    norm_only = timeit(lambda m: ops.rmsnorm_tenant(x, nw, 1e-5), sets)
    sep = timeit(lambda m: m(ops.rmsnorm_tenant(x, nw, 1e-5)), sets)
    fused = timeit(lambda m: m.forward_fused(x, nw, 1e-5), sets)
    line = f"T={T} {name:8s} {mb:6.1f} MB | plain Linear {plain:6.1f} us | rmsnorm alone {norm_only:5.1f} | rmsnorm + Linear {sep:6.1f} | fused norm {fused:6.1f}"
    if il:
        sep3 = timeit(lambda m: ops.swiglu_interleaved8(m(ops.rmsnorm_tenant(x, nw, 1e-5))), sets)
        fused3 = timeit(lambda m: m.forward_fused(x, nw, 1e-5, swiglu=True), sets)
        line += f" | rmsnorm + Linear + swiglu {sep3:6.1f} | fused norm + swiglu {fused3:6.1f}"
    print(line, flush=True)
    del sets
    torch.cuda.empty_cache()
