#!/usr/bin/env python3
"""Timeline of the persistent decode chain (development probe).  Needs a library built with -DBD_CHAIN_TRACE:
    HIPCC_EXTRA=-DBD_CHAIN_TRACE python -m bitdelta_amd.build --force
Prints, per phase, the median over blocks of the owner wave's stamps (us, relative to the block's first stamp)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bitdelta_amd import binary_gemm_kernel as k
from bitdelta_amd._lib import workspace
from bitdelta_amd.serving_loop import FusedDeltaLinear

T, hid, inter, dt = 6, 4096, 14336, torch.float16
g = torch.Generator(device="cuda").manual_seed(0)


def lin(widths, kk, il=False):
    ws = [(torch.randn(n, kk, device="cuda", generator=g) * 0.02).to(dt) for n in widths]
    ms = [torch.randint(-2**31, 2**31 - 1, (T, kk // 32, n), device="cuda", generator=g, dtype=torch.int64).to(torch.int32) for n in widths]
    cs = [torch.rand(T, device="cuda", generator=g) * 1e-3 for _ in widths]
    return FusedDeltaLinear(ws, ms, cs, interleave8=il)


sets = [(lin([hid], hid), lin([inter, inter], hid, il=True), lin([hid], inter), lin([4096, 1024, 1024], hid)) for _ in range(2)]
a = torch.randn(T, 1, hid, device="cuda", generator=g).to(dt)
h = torch.randn(T, 1, hid, device="cuda", generator=g).to(dt)
n1 = torch.ones(T, hid, device="cuda", dtype=dt)
ws, _ = workspace(16384 + (1 << 17), "cuda", zeroed=True)
off = k._CHAIN_SYNC_OFF + 64 + 4096
for it in range(6):
    o, gu, down, qkv = sets[it % 2]
    h_mid, h_out = torch.empty_like(h), torch.empty_like(h)
    act = torch.empty(T, 1, inter, device="cuda", dtype=dt)
    q_out = torch.empty(T, 1, 6144, device="cuda", dtype=dt)
    phases = [dict(x=a, weight=o.weight, mask_packed=o.mask_packed, alpha=o.alpha, out=h_mid, residual=h),
              dict(x=h_mid, weight=gu.weight, mask_packed=gu.mask_packed, alpha=gu.alpha_pair, out=act, norm_weight=n1, eps=1e-5),
              dict(x=act, weight=down.weight, mask_packed=down.mask_packed, alpha=down.alpha, out=h_out, residual=h_mid),
              dict(x=h_out, weight=qkv.weight, mask_packed=qkv.mask_packed, alpha=qkv.alpha, out=q_out, norm_weight=n1, eps=1e-5)]
    k.decode_chain(phases, tenants=T)
    torch.cuda.synchronize()
tr = ws[off:off + 256 * 4 * 6 * 8].view(torch.int64).view(256, 4, 6).cpu().double()
t0 = tr[:, 0, 0].min()
tr = (tr - t0) / 100.0           # s_memtime ticks at 100 MHz -> us
names = ["entry", "first compute", "loop end", "stores acked", "flag published", "barrier passed"]
for ph in range(4):
    print(f"phase {ph}: " + "  ".join(f"{names[i]} med {tr[:, ph, i].median():7.2f} (min {tr[:, ph, i].min():7.2f} max {tr[:, ph, i].max():7.2f})" for i in range(6)))
