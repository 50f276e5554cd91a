#!/usr/bin/env python3
"""Published binary_bmm shapes (M = 1, one mask per row): the one-mask-per-row kernel (variant 800; BD_ROWS_TUNE picks its A/B arms in a child
process) against the kernels it replaced (600 streaming for <= 8 masks, 200 = the round-1 split-k kernel above).  hipGraph of 20 calls,
median of 7 replays / 20, masks rotated so that neither L2 nor the Infinity Cache holds them.

    python tools/bench_rows.py            # runs itself once per BD_ROWS_TUNE arm
"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child():
    import torch
    import bitdelta_amd as bd
    from bitdelta_amd import _lib
    L = _lib.lib()
    tune = os.environ.get("BD_ROWS_TUNE", "0")
    for B, NK in ((8, 4096), (16, 4096), (8, 8192), (16, 8192), (4, 4096), (32, 4096)):
        nset = max(2, int(400e6 // (B * NK * NK // 8)) + 1)
        g = torch.Generator(device="cuda").manual_seed(B + NK)
        xs = [torch.randn(B, 1, NK, device="cuda", generator=g).half() for _ in range(nset)]
        ps = [torch.randint(-2 ** 31, 2 ** 31 - 1, (B, NK // 32, NK), device="cuda", generator=g, dtype=torch.int64).to(torch.int32)
              for _ in range(nset)]
        row = []
        for v in [int(a) for a in os.environ.get('BD_ROWS_VARIANTS', '-1,600,200').split(',')]:
            if v == 600 and B > 8:
                continue
            L.bd_set_gemm_variant(v)
            try:
                for i in range(nset):
                    bd.binary_bmm(xs[i], ps[i])
                ran = L.bd_last_gemm_variant()
                torch.cuda.synchronize()
                s = torch.cuda.Stream()
                with torch.cuda.stream(s):
                    gr = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gr, stream=s):
                        for i in range(20):
                            bd.binary_bmm(xs[i % nset], ps[i % nset])
                    ts = []
                    for _ in range(7):
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record(s); gr.replay(); e1.record(s)
                        s.synchronize()
                        ts.append(e0.elapsed_time(e1) / 20 * 1e3)
                ts.sort()
                us = ts[len(ts) // 2]
                row.append(f"v{v}->{ran}: {us:7.2f} us {B * NK * NK / 8 / us / 1e6:6.2f} TB/s")
            finally:
                L.bd_set_gemm_variant(-1)
        print(f"tune {tune:>2} B={B:2d} N=K={NK} | " + " | ".join(row), flush=True)


def child_shared():
    """one mask shared by M rows (binary_matmul at M <= 16) and a few rows per mask: 800 against the streaming kernel"""
    import torch
    import bitdelta_amd as bd
    from bitdelta_amd import _lib
    L = _lib.lib()
    for B, M, NK in ((1, 1, 4096), (1, 1, 8192), (1, 4, 4096), (1, 8, 4096), (1, 8, 8192), (1, 16, 4096), (1, 16, 8192), (4, 4, 4096), (2, 8, 8192)):
        nset = max(2, int(400e6 // (B * NK * NK // 8)) + 1)
        nset = min(nset, 24)
        g = torch.Generator(device="cuda").manual_seed(B + NK + M)
        xs = [torch.randn(B, M, NK, device="cuda", generator=g).half() for _ in range(nset)]
        ps = [torch.randint(-2 ** 31, 2 ** 31 - 1, (B, NK // 32, NK), device="cuda", generator=g, dtype=torch.int64).to(torch.int32)
              for _ in range(nset)]
        row = []
        for v in (-1, 600, 800):
            L.bd_set_gemm_variant(v)
            try:
                for i in range(nset):
                    bd.binary_bmm(xs[i], ps[i])
                ran = L.bd_last_gemm_variant()
                torch.cuda.synchronize()
                s = torch.cuda.Stream()
                with torch.cuda.stream(s):
                    gr = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gr, stream=s):
                        for i in range(20):
                            bd.binary_bmm(xs[i % nset], ps[i % nset])
                    ts = []
                    for _ in range(7):
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record(s); gr.replay(); e1.record(s)
                        s.synchronize()
                        ts.append(e0.elapsed_time(e1) / 20 * 1e3)
                ts.sort()
                row.append(f"v{v}->{ran}: {ts[len(ts) // 2]:7.2f} us")
            except Exception as e:          # a forced variant outside its envelope
                row.append(f"v{v}: {type(e).__name__}")
            finally:
                L.bd_set_gemm_variant(-1)
        print(f"tune {os.environ.get('BD_ROWS_TUNE', '0'):>2} B={B:2d} M={M:2d} N=K={NK} | " + " | ".join(row), flush=True)


if __name__ == "__main__":
    if os.environ.get("BD_ROWS_CHILD") == "shared":
        child_shared()
    elif os.environ.get("BD_ROWS_CHILD"):
        child()
    else:
        for tune in sys.argv[1:] or ["0", "1", "8", "2", "4", "6"]:
            env = dict(os.environ, BD_ROWS_TUNE=tune, BD_ROWS_CHILD="1")
            subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, check=False, timeout=600)
