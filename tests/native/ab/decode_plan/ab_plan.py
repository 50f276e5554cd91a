"""A/B of the decode plan (one persistent launch, bitdelta_amd/plan.py) against the separate launches of the same decoder layers.

    python tests/native/ab/decode_plan/ab_plan.py [--model mistral-7b] [--tenants 6] [--layers 8] [--kv 512] [--iters 200]

Both legs run the serving loop's decode-layer sequence on the same static buffers:
    q|k|v = Linear(RMSNorm(x)) -> RoPE + KV append + attention -> x += o(a) -> act = SwiGLU(gate|up(RMSNorm(x))) -> x += down(act)
 * separate: 5 HIP launches per layer (norm-fused q|k|v, decode attention, o + residual, norm-fused gate|up + SwiGLU, down + residual),
   replayed from a hipGraph;
 * plan: the same 5 phases per layer in ONE launch, replayed from a hipGraph.
Checks that the final hidden state, the attention output and the KV rows of the two legs are IDENTICAL (bit for bit), then times both.
`--linear-only` drops the attention phase from both legs (the three / four Linear launches alone).
Prints one JSON line per leg.
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", "..", "..")))
import bitdelta_amd as bd                                                   # noqa: E402
from bitdelta_amd import serving_ops as ops                                 # noqa: E402
from bitdelta_amd._lib import lib                                           # noqa: E402
from bitdelta_amd.plan import DecodePlan                                    # noqa: E402
from bitdelta_amd.serving_loop import MODEL_CONFIGS, TenantDecoder          # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="mistral-7b")
    ap.add_argument("--tenants", type=int, default=6)
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--kv", type=int, default=512)
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--linear-only", action="store_true")
    ap.add_argument("--rotate", action="store_true", help="plan leg: one buffer per (layer, phase output) instead of in-place reuse")
    ap.add_argument("--plain", action="store_true", help="plan leg: plain (L2-cached) reads of earlier phases' outputs; implies --rotate")
    ap.add_argument("--barriers", type=int, default=0, help="also time a plan of N empty phases (the device-wide barrier alone)")
    ap.add_argument("--pair", action="store_true", help="third leg: only [attention -> o] of every layer as a 2-phase plan, the rest as launches")
    ap.add_argument("--dtype", default="float16")
    a = ap.parse_args()
    a.rotate = a.rotate or a.plain
    dev = torch.device("cuda:0")
    dtype = getattr(torch, a.dtype)
    T = a.tenants
    hid, inter, _, heads, kvh, _ = MODEL_CONFIGS[a.model]
    hd = hid // heads
    Lc = max(256, a.kv + 64)
    dec = TenantDecoder.synthetic(a.model, T, dev, dtype=dtype, layers=a.layers, max_len=Lc)
    eps = dec.eps
    g = torch.Generator(device=dev).manual_seed(5)
    x0 = (torch.randn(T, 1, hid, device=dev, generator=g) * 0.5).to(dtype)
    kc = [(torch.randn(T, kvh, Lc, hd, device=dev, generator=g) * 0.5).to(dtype) for _ in range(a.layers)]
    vc = [(torch.randn(T, kvh, Lc, hd, device=dev, generator=g) * 0.5).to(dtype) for _ in range(a.layers)]
    valid0 = torch.zeros(T, Lc, dtype=torch.bool, device=dev)
    valid0[:, :a.kv] = True
    valid0[0, :7] = False                                                     # some left padding
    pos = torch.full((1,), a.kv, dtype=torch.int64, device=dev)

    def fresh_state():
        return {"x": x0.clone(), "k": [t.clone() for t in kc], "v": [t.clone() for t in vc], "valid": valid0.clone()}

    # ---------------- separate launches
    def step_sep(st, keep=None):
        x = st["x"]
        for li, layer in enumerate(dec.layers):
            qkv = None
            if not a.linear_only:
                qkv = layer.qkv.forward_fused(x, layer.norm1, eps)
                att = ops.decode_attention(qkv, dec.cos, dec.sin, st["k"][li], st["v"][li], st["valid"], pos, heads, kvh)
            else:
                att = st["att"]
            x = layer.o(att, residual=x)
            x1 = x.clone() if keep is not None else None
            act = layer.gate_up.forward_fused(x, layer.norm2, eps, swiglu=True)
            x = layer.down(act, residual=x)
            if keep is not None:
                keep.append({"qkv": qkv, "att": att.clone(), "x1": x1, "act": act.clone(), "x2": x.clone()})
        return x

    # ---------------- plan
    def build_plan(st):
        t_pad = dec.layers[0].qkv.mask_packed.shape[4]
        plan = DecodePlan(dev, dtype, t_pad, plain_reads=a.plain)
        mk = lambda n: torch.empty(T, 1, n, device=dev, dtype=dtype)
        shared = {"qkv": mk((heads + 2 * kvh) * hd), "att": mk(heads * hd), "act": mk(inter)}
        need = lib().bd_srv_decode_attention_workspace_bytes(T, heads, kvh, hd, Lc)
        ws = torch.zeros(max(int(need), 16), dtype=torch.uint8, device=dev) if need > 0 else None
        x = st["x"]
        per_layer = []
        for li, layer in enumerate(dec.layers):
            b = {k: (mk(v.shape[2]) if a.rotate else v) for k, v in shared.items()}
            if a.linear_only:
                b["att"] = st["att"]
            else:
                plan.linear(x, layer.qkv.weight_tiled, layer.qkv.mask_packed, layer.qkv.alpha, b["qkv"], groups=layer.qkv.groups,
                            norm_weight=layer.norm1, eps=eps)
                plan.attention(b["qkv"], dec.cos, dec.sin, st["k"][li], st["v"][li], st["valid"], pos, b["att"], heads, kvh, ws)
            if a.rotate:
                b["x1"], b["x2"] = mk(hid), mk(hid)
                plan.linear(b["att"], layer.o.weight_tiled, layer.o.mask_packed, layer.o.alpha, b["x1"], groups=layer.o.groups, residual=x)
                plan.linear(b["x1"], layer.gate_up.weight_tiled, layer.gate_up.mask_packed, layer.gate_up.alpha_pair, b["act"], groups=2,
                            norm_weight=layer.norm2, eps=eps, swiglu=True)
                plan.linear(b["act"], layer.down.weight_tiled, layer.down.mask_packed, layer.down.alpha, b["x2"], groups=layer.down.groups,
                            residual=b["x1"])
                x = b["x2"]
            else:
                plan.linear(b["att"], layer.o.weight_tiled, layer.o.mask_packed, layer.o.alpha, x, groups=layer.o.groups, accumulate=True)
                plan.linear(x, layer.gate_up.weight_tiled, layer.gate_up.mask_packed, layer.gate_up.alpha_pair, b["act"], groups=2,
                            norm_weight=layer.norm2, eps=eps, swiglu=True)
                plan.linear(b["act"], layer.down.weight_tiled, layer.down.mask_packed, layer.down.alpha, x, groups=layer.down.groups,
                            accumulate=True)
            per_layer.append(b)
        return plan.finalize(), per_layer, x

    att_fixed = (torch.randn(T, 1, heads * hd, device=dev, generator=g) * 0.5).to(dtype)

    # ---- correctness: same initial state, one step each
    s1 = fresh_state(); s1["att"] = att_fixed
    keep = []
    x_sep = step_sep(s1, keep).clone()
    s2 = fresh_state(); s2["att"] = att_fixed.clone()
    plan, per_layer, x_out = build_plan(s2)
    plan.launch()
    torch.cuda.synchronize()
    plan.check_status()
    same_kv = a.linear_only or all(torch.equal(s1["k"][i], s2["k"][i]) and torch.equal(s1["v"][i], s2["v"][i]) for i in range(a.layers))
    report = {"check": "plan == separate launches", "x": torch.equal(x_sep, x_out), "kv": same_kv, "valid": torch.equal(s1["valid"], s2["valid"]),
              "x_mismatches": int((x_sep != x_out).sum()), "finite": bool(torch.isfinite(x_out.float()).all()), "phases": len(plan),
              "layers": a.layers, "rotate": a.rotate, "plain_reads": a.plain}
    if a.rotate:                       # every intermediate of every layer survives the launch: name the first phase that differs
        first_bad = None
        for li in range(a.layers):
            for k in ("qkv", "att", "x1", "act", "x2"):
                if keep[li][k] is None or (a.linear_only and k in ("qkv", "att")):
                    continue
                if not torch.equal(keep[li][k], per_layer[li][k]) and first_bad is None:
                    first_bad = "layer %d %s (%d of %d elements)" % (li, k, int((keep[li][k] != per_layer[li][k]).sum()), keep[li][k].numel())
        report["first_mismatch"] = first_bad
    else:
        report["act_last"] = torch.equal(keep[-1]["act"], per_layer[-1]["act"])
        report["att_last"] = a.linear_only or torch.equal(keep[-1]["att"], per_layer[-1]["att"])
    print(json.dumps(report), flush=True)

    # a second launch of the same plan on a DIFFERENT input (the barrier words must have been left at zero; with plain reads, no line of
    # the first launch's intermediates may survive in an L2)
    x0b = (torch.randn(T, 1, hid, device=dev, generator=g) * 0.5).to(dtype)
    s1b = fresh_state(); s1b["att"] = att_fixed; s1b["x"].copy_(x0b)
    x_sep_b = step_sep(s1b).clone()
    s2["x"].copy_(x0b)
    for i in range(a.layers):
        s2["k"][i].copy_(kc[i]); s2["v"][i].copy_(vc[i])
    s2["valid"].copy_(valid0)
    plan.launch()
    torch.cuda.synchronize()
    plan.check_status()
    print(json.dumps({"check": "second launch of the plan, new input", "x": torch.equal(x_sep_b, x_out),
                      "x_mismatches": int((x_sep_b != x_out).sum())}), flush=True)

    # ---- timing (hipGraph replays; x drifts over replays, which the kernels do not care about)
    def bench(fn, iters):
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            fn()
        torch.cuda.current_stream(dev).wait_stream(side)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=side):
            fn()
        torch.cuda.synchronize()
        for _ in range(5):
            gr.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            gr.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / iters

    s3 = fresh_state(); s3["att"] = att_fixed.clone()

    def sep_fn():
        step_sep(s3)

    us_sep = bench(sep_fn, a.iters)
    us_plan = bench(plan.launch, a.iters)
    plan.check_status()
    # after a few hundred replays: new input once more
    x0c = (torch.randn(T, 1, hid, device=dev, generator=g) * 0.5).to(dtype)
    s1c = fresh_state(); s1c["att"] = att_fixed; s1c["x"].copy_(x0c)
    x_sep_c = step_sep(s1c).clone()
    s2["x"].copy_(x0c)
    for i in range(a.layers):
        s2["k"][i].copy_(kc[i]); s2["v"][i].copy_(vc[i])
    s2["valid"].copy_(valid0)
    plan.launch()
    torch.cuda.synchronize()
    plan.check_status()
    print(json.dumps({"check": "launch after the replays, new input", "x": torch.equal(x_sep_c, x_out),
                      "x_mismatches": int((x_sep_c != x_out).sum())}), flush=True)
    if a.pair and not a.linear_only:
        # [attention -> o + residual] as a 2-phase plan per layer (the o projection's weight prologue lands while the attention phase runs);
        # q|k|v, gate|up, down stay launches, written into static buffers through the C ABI
        from bitdelta_amd._lib import DTYPE_CODE, check, ptr, stream_ptr
        L = lib()
        dt = DTYPE_CODE[dtype]

        def launch_into(x, fl, y, alpha, groups, accumulate=False, norm=None, swiglu=False):
            N, K = fl.weight.shape
            if norm is not None or swiglu:
                check(L.bd_binary_linear_decode_fused(ptr(x), ptr(fl.weight_tiled), ptr(fl.mask_packed), fl.mask_packed.shape[4], ptr(alpha), ptr(y),
                                                      T, 1, N, K, x.stride(0), x.stride(1), 0, 1, groups, groups, y.stride(0), y.stride(1), dt, dt,
                                                      1 if accumulate else 0, ptr(norm), norm.stride(0) if norm is not None else 0, float(eps),
                                                      1 if swiglu else 0, stream_ptr()), "decode_fused")
            else:
                check(L.bd_binary_linear_decode(ptr(x), ptr(fl.weight_tiled), ptr(fl.mask_packed), 2, fl.mask_packed.shape[4], ptr(alpha), ptr(y),
                                                T, 1, N, K, x.stride(0), x.stride(1), 0, 1, groups, groups, y.stride(0), y.stride(1), dt, dt,
                                                1 if accumulate else 0, stream_ptr()), "decode")

        s4 = fresh_state()
        mk = lambda n: torch.empty(T, 1, n, device=dev, dtype=dtype)
        qkv_b, att_b, act_b = mk((heads + 2 * kvh) * hd), mk(heads * hd), mk(inter)
        need = lib().bd_srv_decode_attention_workspace_bytes(T, heads, kvh, hd, Lc)
        ws4 = torch.zeros(max(int(need), 16), dtype=torch.uint8, device=dev)
        t_pad = dec.layers[0].qkv.mask_packed.shape[4]
        pair_plans = []
        for li, layer in enumerate(dec.layers):
            pp = DecodePlan(dev, dtype, t_pad)
            pp.attention(qkv_b, dec.cos, dec.sin, s4["k"][li], s4["v"][li], s4["valid"], pos, att_b, heads, kvh, ws4)
            pp.linear(att_b, layer.o.weight_tiled, layer.o.mask_packed, layer.o.alpha, s4["x"], groups=layer.o.groups, accumulate=True)
            pair_plans.append(pp.finalize())

        def pair_fn():
            x = s4["x"]
            for li, layer in enumerate(dec.layers):
                launch_into(x, layer.qkv, qkv_b, layer.qkv.alpha, layer.qkv.groups, norm=layer.norm1)
                pair_plans[li].launch()
                launch_into(x, layer.gate_up, act_b, layer.gate_up.alpha_pair, 2, norm=layer.norm2, swiglu=True)
                launch_into(act_b, layer.down, x, layer.down.alpha, layer.down.groups, accumulate=True)

        pair_fn()
        torch.cuda.synchronize()
        for pp in pair_plans:
            pp.check_status()
        s1p = fresh_state(); s1p["att"] = att_fixed
        x_ref = step_sep(s1p).clone()
        print(json.dumps({"check": "pair plans == separate launches", "x": torch.equal(x_ref, s4["x"]),
                          "x_mismatches": int((x_ref != s4["x"]).sum())}), flush=True)
        us_pair = bench(pair_fn, a.iters)
        for pp in pair_plans:
            pp.check_status()
        print(json.dumps({"leg": "[attention -> o] as a 2-phase plan per layer, other launches separate", "us_per_step": round(us_pair, 2),
                          "us_per_layer": round(us_pair / a.layers, 2), "vs_separate": round(us_pair / us_sep, 4)}), flush=True)
    if a.barriers > 0:
        bp = DecodePlan(dev, dtype, dec.layers[0].qkv.mask_packed.shape[4])
        for _ in range(a.barriers):
            bp.barrier()
        bp.finalize()
        us_b = bench(bp.launch, 50)
        bp.check_status()
        one = DecodePlan(dev, dtype, dec.layers[0].qkv.mask_packed.shape[4])
        one.barrier()
        one.finalize()
        us_1 = bench(one.launch, 50)
        print(json.dumps({"barrier_phases": a.barriers, "us_per_launch": round(us_b, 2), "us_one_phase_launch": round(us_1, 2),
                          "us_per_barrier": round((us_b - us_1) / max(a.barriers - 1, 1), 3)}), flush=True)
    byt = sum(l.o.linear_bytes() + l.gate_up.linear_bytes() + l.down.linear_bytes() + (0 if a.linear_only else l.qkv.linear_bytes())
              for l in dec.layers)
    for name, us in (("separate launches (hipGraph)", us_sep), ("decode plan (one launch)", us_plan)):
        print(json.dumps({"leg": name, "model": a.model, "tenants": T, "layers": a.layers, "kv": a.kv, "linear_only": a.linear_only,
                          "us_per_step": round(us, 2), "us_per_layer": round(us / a.layers, 2),
                          "linear_GBps": round(byt / us * 1e-3, 1)}), flush=True)
    print(json.dumps({"plan_over_separate": round(us_plan / us_sep, 4)}), flush=True)


if __name__ == "__main__":
    main()
