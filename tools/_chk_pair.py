import sys, torch
sys.path.insert(0, '.')
from bitdelta_amd import _lib
from bitdelta_amd.serving_loop import TenantDecoder
L = _lib.lib()
dec = TenantDecoder.synthetic("mistral-7b", 6, "cuda", dtype=torch.float16, seed=1, layers=2, max_len=256)
lay = dec.layers[0]
x = torch.randn(6, 64, 4096, device="cuda", dtype=torch.float16)
def t(fn, n=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, lin, xin, res in (("qkv", lay.qkv, x, None), ("o", lay.o, x, x.clone()), ("gate_up", lay.gate_up, x, None),
                            ("down", lay.down, torch.randn(6, 64, 14336, device="cuda", dtype=torch.float16), x.clone())):
    y = lin(xin, residual=res) if res is not None else lin(xin)
    v = L.bd_last_gemm_variant()
    us = t(lambda: lin(xin, residual=res) if res is not None else lin(xin))
    usn = t(lambda: lin(xin))
    print(name, "variant", v, f"{us:.1f} us (with residual: {res is not None}); without residual {usn:.1f} us variant", (lin(xin), L.bd_last_gemm_variant())[1], "groups", lin.groups, "interleave8", lin.interleave8)
