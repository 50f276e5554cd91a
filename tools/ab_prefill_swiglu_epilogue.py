#!/usr/bin/env python3
"""Same-process A/B of the headline prefill step (Llama-2-7B shapes, 2048 tokens): gate|up GEMM + one SwiGLU pass (shipped) against SwiGLU in the GEMM's
epilogue (bd_binary_linear_swiglu, variant 15).  python tools/ab_prefill_swiglu_epilogue.py [layers]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_model as bm

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda", 0)
dec = bm.Decoder("llama-2-7b", dev, torch.bfloat16, layers=layers, seed=0)
ids = torch.randint(0, 32000, (1, 2048), device=dev)


def timed(reps=5):
    for _ in range(2):
        dec(ids)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = dec(ids)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, out


res = {}
for rnd in range(3):
    for flag in (False, True):
        for l in dec.layers:
            l.swiglu_epilogue = flag
        ms, out = timed()
        res.setdefault(flag, []).append(ms)
        if rnd == 0:
            res[("out", flag)] = out.clone()
print(f"{layers} layers, 2048 tokens: GEMM + SwiGLU pass {sorted(res[False])[1]:.3f} ms | SwiGLU in the epilogue {sorted(res[True])[1]:.3f} ms "
      f"(median of 3 x 5 steps); logits identical: {torch.equal(res[('out', False)], res[('out', True)])}")
