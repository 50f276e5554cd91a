#!/usr/bin/env python3
"""Same-process A/B of the multi-tenant decode step (hipGraph replay) under library tuning flags / serving-loop switches.

    python tools/ab_decode_step.py --tenants 6 --arms base:0 fg_off:256 fg_all:512 pf:0:prefetch

An arm is name:stream_tuning_flags[:prefetch|nostep].  Each arm is captured as its own graph (dispatch decisions are taken at capture time);
the arms are then timed alternately, `--rounds` rounds of `--steps` replays each; min and median per arm are printed.
Replaces the one-off tools/gpu_r4*.sh / gpu_r5*.sh scripts of earlier rounds for this kind of question."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="mistral-7b")
    ap.add_argument("--tenants", type=int, default=6)
    ap.add_argument("--kv-len", type=int, default=512)
    ap.add_argument("--layers", type=int, default=None)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--arms", nargs="+", default=["base:0"])
    args = ap.parse_args()
    from bitdelta_amd import _lib, dist as bdd
    from bitdelta_amd.serving_loop import TenantDecoder
    dev = torch.device("cuda", 0)
    L = _lib.lib()
    T = args.tenants
    dec = TenantDecoder.synthetic(args.model, T, dev, dtype=torch.float16, seed=4321, layers=args.layers, max_len=args.kv_len + 256)
    vocab = dec.cfg[5]
    g = torch.Generator().manual_seed(4321)
    prompts = [torch.randint(1, vocab, (args.kv_len,), generator=g).tolist() for _ in range(T)]
    ids, am = dec.prepare(prompts)
    cache = dec.new_cache()
    first = torch.argmax(dec.prefill(ids, am, cache), dim=-1)
    st = {"cache": cache, "tok": first[:, None].clone(), "pos": torch.tensor([ids.shape[1]], device=dev),
          "step": torch.tensor([1], device=dev), "stop_ids": torch.full((T, 1), -1, dtype=torch.long, device=dev),
          "out": torch.zeros(T, 4096, dtype=torch.long, device=dev), "stopped": torch.zeros(T, dtype=torch.bool, device=dev)}
    snap = {k: v.clone() for k, v in st.items() if torch.is_tensor(v)}
    valid0 = cache["valid"].clone()

    def restore():
        for k, v in snap.items():
            st[k].copy_(v)
        cache["valid"].copy_(valid0)

    runners, toks = {}, {}
    for arm in args.arms:
        parts = arm.split(":")
        name, flags = parts[0], int(parts[1]) if len(parts) > 1 else 0
        dec.prefetch_o = len(parts) > 2 and "prefetch" in parts[2]
        dec.step_kernels = not (len(parts) > 2 and "nostep" in parts[2])          # (stock torch ops at both ends of the step)
        L.bd_set_stream_tuning(flags)
        restore()
        runners[name] = dec._graph_runner(st)
        # the tokens the arm produces over 8 steps: every arm must agree (the switches change no arithmetic beyond summation forms)
        restore()
        for _ in range(8):
            runners[name]()
        torch.cuda.synchronize()
        toks[name] = st["out"][:, 1:9].cpu().clone()
    L.bd_set_stream_tuning(0)
    dec.prefetch_o = False
    dec.step_kernels = True
    ms = {n: [] for n in runners}
    for n, run in runners.items():          # warm-up
        restore()
        for _ in range(12):
            run()
    for _ in range(args.rounds):
        for n, run in runners.items():
            restore()
            run()
            ms[n].append(bdd.timed_region(run, args.steps, device_sync=torch.cuda.synchronize) / args.steps * 1e3)
    names = list(runners)
    out = {"model": args.model, "tenants": T, "kv_len": args.kv_len, "layers": len(dec.layers), "steps": args.steps,
           "arms": {n: {"min_ms": min(v), "median_ms": sorted(v)[len(v) // 2], "all_ms": [round(x, 4) for x in v],
                        "tokens_equal_first_arm": bool(torch.equal(toks[n], toks[names[0]]))} for n, v in ms.items()}}
    print(json.dumps(out))
    for n in names:
        a = out["arms"][n]
        print(f"# {n:28s} min {a['min_ms']:.4f}  median {a['median_ms']:.4f} ms/step  tokens==first arm: {a['tokens_equal_first_arm']}", file=sys.stderr)


if __name__ == "__main__":
    main()
