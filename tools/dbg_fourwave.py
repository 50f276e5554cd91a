"""Reproduce helper: the fused + residual check of test_four_wave_persistent_kernels for many residual draws; prints the worst violation."""
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch
import bitdelta_amd as bd
from bitdelta_amd import _lib
from oracle import bd_oracle as oracle
import test_gpu_parity as T
L = _lib.lib()
bad = 0
for dt in (torch.bfloat16, torch.float16):
    a, p, w, al1 = T.rand_problem(2, 300, 256, 512, dt, 2, seed=23)
    y32 = oracle.binary_linear(a, w, p, al1, out_dtype=torch.float32)
    for trial in range(40):
        torch.manual_seed(1000 + trial)
        r = torch.randn(2, 300, 512).to(dt)
        L.bd_set_gemm_variant(14)
        got = bd.binary_linear(a.cuda(), w.cuda(), p.cuda(), al1.cuda(), residual=r.cuda().clone())
        y16 = bd.binary_linear(a.cuda(), w.cuda(), p.cuda(), al1.cuda())
        L.bd_set_gemm_variant(-1)
        want = (r.float() + y32.to(dt).float()).to(dt)
        d = (got.cpu().float() - want.float()).abs()
        tol = (r.float().abs() + y32.abs()) * (2 ** -10 if dt == torch.float16 else 2 ** -7) + 1e-4
        viol = (d > tol)
        eq = torch.equal(got, r.cuda() + y16)
        if viol.any() or not eq:
            bad += 1
            idx = viol.nonzero()[:5].tolist()
            print(dt, trial, "violations", int(viol.sum()), "equal-to-separate", eq, "first", idx, [(float(got.cpu()[tuple(i)]), float(want[tuple(i)]), float(r[tuple(i)]), float(y32[tuple(i)])) for i in idx][:3])
print("bad trials", bad)
