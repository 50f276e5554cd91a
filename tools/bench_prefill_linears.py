#!/usr/bin/env python3
"""Per-launch timing of a Llama-2-7B layer's projections at prefill (M = 2048): the seven separate BinaryDiff launches vs the
four fused ones (q|k|v, o + residual, gate|up -> SwiGLU, down + residual).  Back-to-back launches, HIP events around each batch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bitdelta_amd as bd
from bitdelta_amd import _lib, serving_ops as ops
from bitdelta_amd.binary_gemm_kernel import binary_linear, binary_linear_swiglu
from bitdelta_amd.serving_loop import FusedDeltaLinear
from bitdelta_amd.diff import binarize

dev = "cuda"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
hid, inter = 4096, 11008
g = torch.Generator(device=dev).manual_seed(0)


def pair(n, k):
    w = (torch.randn(n, k, device=dev, generator=g) * 0.02).bfloat16()
    f = (w.float() + torch.randn(n, k, device=dev, generator=g) * 5e-4).bfloat16()
    m, c = binarize(w, f)
    return w, m[None].contiguous(), c.reshape(1, 1).float()


def timeit(name, fn, flops, iters=40):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    print(f"  {name:46s} {us:9.1f} us   {flops / us * 1e-6:7.1f} TF   variant {_lib.lib().bd_last_gemm_variant()}")
    return us


x = torch.randn(1, M, hid, device=dev, generator=g).bfloat16()
xi = torch.randn(1, M, inter, device=dev, generator=g).bfloat16()
res = torch.randn(1, M, hid, device=dev, generator=g).bfloat16()
q, k, v, o = pair(hid, hid), pair(hid, hid), pair(hid, hid), pair(hid, hid)
ga, up, dn = pair(inter, hid), pair(inter, hid), pair(hid, inter)
qkv = FusedDeltaLinear([q[0], k[0], v[0]], [q[1], k[1], v[1]], [q[2].reshape(1), k[2].reshape(1), v[2].reshape(1)], decode_copies=False)
gu = FusedDeltaLinear([ga[0], up[0]], [ga[1], up[1]], [ga[2].reshape(1), up[2].reshape(1)], interleave8=True, decode_copies=False)
print(f"M = {M}")
t7 = 0
t7 += 3 * timeit("q / k / v  (each, 4096x4096)", lambda: binary_linear(x, *q), 4.0 * M * hid * hid)
t7 += timeit("o + residual", lambda: binary_linear(x, *o, residual=res), 4.0 * M * hid * hid)
t7 += 2 * timeit("gate / up (each, 11008x4096)", lambda: binary_linear(x, *ga), 4.0 * M * hid * inter)
gg, uu = binary_linear(x, *ga), binary_linear(x, *up)
t7 += timeit("swiglu2 (separate launch)", lambda: ops.swiglu2(gg, uu), 0.0)
t7 += timeit("down + residual (4096x11008)", lambda: binary_linear(xi, *dn, residual=res), 4.0 * M * hid * inter)
t4 = 0
t4 += timeit("q|k|v fused (12288x4096)", lambda: qkv(x), 12.0 * M * hid * hid)
t4 += timeit("o + residual", lambda: binary_linear(x, *o, residual=res), 4.0 * M * hid * hid)
t4 += timeit("gate|up -> SwiGLU fused (22016x4096)", lambda: gu.forward_swiglu(x), 8.0 * M * hid * inter)
timeit("gate|up fused, no epilogue (22016x4096)", lambda: gu(x), 8.0 * M * hid * inter)
t4 += timeit("down + residual (4096x11008)", lambda: binary_linear(xi, *dn, residual=res), 4.0 * M * hid * inter)
fl = 4.0 * M * (4 * hid * hid + 3 * hid * inter)
print(f"  per layer: 7 launches + swiglu {t7:.1f} us ({fl / t7 * 1e-6:.1f} TF)   4 launches {t4:.1f} us ({fl / t4 * 1e-6:.1f} TF)")
L = _lib.lib()
for ts in (0, 1):
    L.bd_set_tail_split(ts)
    timeit(f"gate|up fused, tail split {ts}", lambda: gu(x), 8.0 * M * hid * inter, iters=200)
    timeit(f"  + swiglu_interleaved8", lambda: ops.swiglu_interleaved8(gu(x)), 8.0 * M * hid * inter, iters=200)
    timeit(f"q|k|v fused, tail split {ts}", lambda: qkv(x), 12.0 * M * hid * hid, iters=200)
L.bd_set_tail_split(1)
