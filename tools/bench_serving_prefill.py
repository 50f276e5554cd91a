#!/usr/bin/env python3
"""Prefill latency of the serving loop (TenantDecoder.prefill: T left-padded prompts through one base + T 1-bit deltas), with the HIP
prefill attention (in-place RoPE + bd_srv_prefill_attention) and with the stock path (rope + SDPA over the [T, 1, L, Lc] mask).

    python tools/bench_serving_prefill.py [--model mistral-7b --tenants 6 --layers 32 --lens 64,256,1024]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="mistral-7b")
    ap.add_argument("--tenants", type=int, default=6)
    ap.add_argument("--layers", type=int, default=None)
    ap.add_argument("--lens", default="64,256,1024")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--modes", default="torch,hip,fused", help="torch = rope + SDPA(mask); hip = HIP RoPE + attention; fused = + the round-6 short-prompt fusions; BD_SWIGLU_EPI=1: SwiGLU in the pair-tile epilogue too")
    args = ap.parse_args()
    from bitdelta_amd.serving_loop import TenantDecoder
    dec = TenantDecoder.synthetic(args.model, args.tenants, "cuda", dtype=torch.bfloat16, seed=1, layers=args.layers, shared_heads=True)
    g = torch.Generator().manual_seed(0)
    for L in [int(v) for v in args.lens.split(",")]:
        lens = [max(1, L - 7 * t) for t in range(args.tenants)]                  # uneven prompts, padded to L
        prompts = [torch.randint(1, 30000, (n,), generator=g).tolist() for n in lens]
        ids, am = dec.prepare(prompts)
        row = []
        modes = args.modes.split(",")
        for flag, fus in ((False, True), (True, False), (True, True)):
            if ("torch", "hip", "fused")[flag + (flag and fus)] not in modes:
                row.append(float("nan"))
                continue
            dec.hip_prefill_attention = flag
            dec.short_prompt_fusions = fus
            dec.swiglu_epilogue = fus and os.environ.get("BD_SWIGLU_EPI", "0") == "1"
            cache = dec.new_cache(ids.shape[1] + 8)
            for _ in range(2):
                dec.prefill(ids, am, cache)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                dec.prefill(ids, am, cache)
            e1.record()
            torch.cuda.synchronize()
            row.append(e0.elapsed_time(e1) / args.reps)
        toks = sum(lens)
        print(f"{args.model}, {args.tenants} tenants, prompts padded to {ids.shape[1]} ({toks} real tokens): prefill {row[0]:.2f} ms with torch "
              f"rope + SDPA(mask), {row[1]:.2f} ms with the HIP RoPE + attention kernels, {row[2]:.2f} ms with the round-6 short-prompt fusions too "
              f"(RoPE + cache append in one launch, the norms on the split-k reduce launches; {toks / row[2] * 1e3:.0f} prompt tokens/s)")


if __name__ == "__main__":
    main()
