#!/usr/bin/env python3
"""Is the headline prefill step (Llama-2-7B shapes, 2048 tokens) host-bound anywhere?  Eager launches against a hipGraph replay of the same step.
python tools/ab_prefill_graph.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_model as bm

dev = torch.device("cuda", 0)
dec = bm.Decoder("llama-2-7b", dev, torch.bfloat16, seed=0)
ids = torch.randint(0, 32000, (1, 2048), device=dev)


def timed(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


eager = sorted(timed(lambda: dec(ids)) for _ in range(3))[1]
side = torch.cuda.Stream(device=dev)
side.wait_stream(torch.cuda.current_stream(dev))
with torch.cuda.stream(side):
    dec(ids)
torch.cuda.current_stream(dev).wait_stream(side)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side):
    out = dec(ids)
torch.cuda.synchronize()
graph = sorted(timed(g.replay) for _ in range(3))[1]
print(f"prefill 2048 tokens: eager {eager:.3f} ms | hipGraph replay {graph:.3f} ms")
