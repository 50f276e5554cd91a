#!/usr/bin/env python3
"""Decode step of ONE rank's Llama-2-70B TP = 8 shards (exchange stubbed: single process) for a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d <out> -o t -- python tools/trace_tp_shard.py [layers]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bitdelta_amd.tp import TPDecoder, LLAMA_70B
from bitdelta_amd import dist as bdd

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0)
dec = TPDecoder(LLAMA_70B, dev, torch.bfloat16, 0, 8, layers=layers, seed=77, max_len=1024)
ids = torch.randint(0, 32000, (1, 512), device=dev)
cache = dec.new_cache(1024)
dec(ids, torch.arange(512, device=dev), cache)
tok, pos = ids[:, :1].clone(), torch.tensor([512], device=dev)
run, _ = dec.decode_runner(tok, pos, cache, use_graph=True)
def step():
    pos.fill_(512); run()
for _ in range(5): step()
t = bdd.timed_region(step, 20, device_sync=torch.cuda.synchronize) / 20
print(f"tp70b shard decode, {layers} layers: {t * 1e3:.3f} ms/step = {t * 1e6 / layers:.1f} us per layer")
