import torch, sys, time
sys.path.insert(0, "/root/repo")
import bitdelta_amd as bd
from bitdelta_amd import _lib
L = _lib.lib()
dev = "cuda"
M = N = K = 4096
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(1, M, K, device=dev, generator=g).bfloat16()
p = torch.randint(-2 ** 31, 2 ** 31 - 1, (1, K // 32, N), device=dev, generator=g, dtype=torch.int64).to(torch.int32)
out = torch.empty(1, M, N, device=dev, dtype=torch.bfloat16)
def run(v, iters=200):
    L.bd_set_gemm_variant(v)
    for _ in range(20): bd.delta_bmm(x, p, out=out, round_mode=0)
    torch.cuda.synchronize()
    lv = L.bd_last_gemm_variant()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): bd.delta_bmm(x, p, out=out, round_mode=0)
    e1.record(); torch.cuda.synchronize()
    t_all = e0.elapsed_time(e1) / iters
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
    for a, b in ev:
        a.record(); bd.delta_bmm(x, p, out=out, round_mode=0); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    L.bd_set_gemm_variant(-1)
    print(f"forced {v}: ran {lv}: back-to-back {t_all*1e3:.2f} us/launch; per-launch events avg {sum(ts)/len(ts)*1e3:.2f} med {ts[15]*1e3:.2f} min {ts[0]*1e3:.2f}")
for v in (-1, 13, 0, 13, 0, -1):
    run(v)
