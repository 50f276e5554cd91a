#!/usr/bin/env python3
"""Experiment: tile-major base weight for the streaming decode kernel (gemv_stream_kernel WT = 1) vs the reference row-major [N, K],
cold weights, same process.  Checks bit-identity first.  usage: python tools/bench_tiled_w.py [T]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bitdelta_amd._lib import lib
from bitdelta_amd.binary_gemm_kernel import binary_linear_decode, pack_decode_masks

T = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev, dt = "cuda", torch.float16
g = torch.Generator(device=dev).manual_seed(0)


def tile_weight(w):
    N, K = w.shape
    assert N % 16 == 0 and K % 128 == 0
    return w.view(N // 16, 16, K // 128, 4, 4, 8).permute(0, 2, 3, 1, 4, 5).contiguous().view(N, K)


def timeit(fn, n, reps=40):
    for i in range(n):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(i % n)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


L = lib()
for name, N, K, nset in (("o", 4096, 4096, 14), ("q+k+v", 6144, 4096, 9), ("gate+up", 28672, 4096, 2), ("down", 4096, 14336, 4)):
    sets = []
    for _ in range(nset):
        w = (torch.randn(N, K, device=dev, generator=g) * 0.02).to(dt)
        m = torch.randint(-2**31, 2**31 - 1, (T, K // 32, N), device=dev, generator=g, dtype=torch.int64).to(torch.int32)
        sets.append((w, tile_weight(w), pack_decode_masks(m)))
        del m
    al = torch.rand(T, 1, device=dev, generator=g) * 1e-3
    x = torch.randn(T, 1, K, device=dev, generator=g).to(dt)
    w, wt, pk = sets[0]
    ref = binary_linear_decode(x, w, pk, al, layout="packed")
    L.bd_set_stream_tuning(32)
    got = binary_linear_decode(x, wt, pk, al, layout="packed")
    L.bd_set_stream_tuning(0)
    same = torch.equal(ref, got)
    graphs = {}
    for mode in (0, 32):
        L.bd_set_stream_tuning(mode)
        gs = []
        for (w, wt, pk) in sets:                       # one graph per weight set: launch overhead out of the picture
            ww = wt if mode else w
            binary_linear_decode(x, ww, pk, al, layout="packed")
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                for _ in range(1):
                    binary_linear_decode(x, ww, pk, al, layout="packed")
            gs.append(gr)
        graphs[mode] = gs
    L.bd_set_stream_tuning(0)
    res = {}
    for rep in range(2):
        for mode in (0, 32):
            res.setdefault(mode, []).append(timeit(lambda i: graphs[mode][i].replay(), nset))
    mb = (2.0 * N * K + T * N * K / 8) / 1e6
    print(f"T={T} {name:8s} {mb:6.1f} MB  bit-identical={same}  row-major {min(res[0]):6.1f} us ({mb / min(res[0]):.2f} TB/s)   "
          f"tile-major {min(res[32]):6.1f} us ({mb / min(res[32]):.2f} TB/s)", flush=True)
    del sets, graphs
    torch.cuda.empty_cache()
