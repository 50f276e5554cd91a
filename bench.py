#!/usr/bin/env python3
"""bench.py -- headline benchmark of the 1-bit-delta Linear hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

Workload (BASELINE.json configs[1]): Llama-2-7B base + one 1-bit delta (Vicuna-7B-v1.5 shapes), prefill of one
2048-token sequence per GPU, synthetic weights/activations (SURVEY.md 8d recipe).  A "step" is one full prefill
forward: 32 layers x 7 fused BinaryDiff projections (the hot path, hand-written HIP) + attention/norm/embedding/lm_head
(stock torch, the callers of the path).  N GPUs = N independent replicas (weak scaling, no data-path collective).

Prints ONE JSON line (rank 0) with the driver contract fields plus
  roofline      the dominant kernel (fused base+delta MFMA GEMM): algorithmic FLOPs of its launches / their summed
                durations, measured live with HIP events on the launch stream inside the timed region
  delta_gemm    the W1A16 delta-GEMM alone at 4096x4096, M = 4096 (the north star's 70 %-of-peak target), same method
  cpu_baseline  the reference's CPU-executable form of the same projections (oracle/torch_port.py), bounded sample
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0        # dense MFMA peak, MI355X_MICROARCH.md (AMD's 5 PF figure is 2:1 sparse)
PEAK_HBM_GBS = 8000.0


class LaunchTimer:
    """Wraps bitdelta_amd's binary_linear so every launch inside the timed region is bracketed by HIP events recorded on
    the stream the kernel is launched on (torch's current stream is the one handed to the C ABI)."""

    def __init__(self):
        self.records = []
        self.enabled = False

    def install(self):
        import bitdelta_amd.binary_gemm_kernel as k
        import bitdelta_amd.diff as d
        import bitdelta_amd.serving as s
        orig = k.binary_linear
        timer = self

        def timed(x, weight, mask, alpha, **kw):
            if not timer.enabled:
                return orig(x, weight, mask, alpha, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            y = orig(x, weight, mask, alpha, **kw)
            e1.record()
            B, M, K = x.shape
            N = weight.shape[0]
            nbytes = 2.0 * B * M * K + 2.0 * N * K + mask.shape[0] * K * N / 8.0 + 4.0 * mask.shape[0] + 2.0 * B * M * N
            timer.records.append((e0, e1, 4.0 * B * M * K * N, nbytes))
            return y
        k.binary_linear = d.binary_linear = s.binary_linear = timed

    def summary(self):
        ms = sum(a.elapsed_time(b) for a, b, _, _ in self.records)
        fl = sum(f for _, _, f, _ in self.records)
        by = sum(n for _, _, _, n in self.records)
        return len(self.records), ms, fl, by


def cpu_baseline(seq=128):
    """Reference-equivalent CPU path (oracle/torch_port.py) for ONE decoder layer's 7 projections (Llama-2-7B shapes) at
    `seq` tokens, all host threads; extrapolated x32 layers to tokens/s.  Attention/norms are excluded (GPU side: <5 %)."""
    from oracle import torch_port as tp
    torch.manual_seed(0)
    hid, inter = 4096, 11008
    shapes = [(hid, hid)] * 4 + [(inter, hid)] * 2 + [(hid, inter)]
    layers = []
    for n_out, n_in in shapes:
        w = (torch.randn(n_out, n_in) * 0.02).bfloat16()
        mask = torch.randint(-2 ** 31, 2 ** 31 - 1, (n_in // 32, n_out), dtype=torch.int64).to(torch.int32)
        layers.append((w, mask, torch.tensor(4e-4)))
    xs = {n_in: torch.randn(1, seq, n_in).bfloat16() for n_in in (hid, inter)}
    def one_layer():
        for w, mask, c in layers:
            tp.forward_unpack_in_loop(xs[w.shape[1]], w, mask, c)
    one_layer()                                   # warm-up
    t0 = time.perf_counter()
    reps = 0
    while reps < 3 or time.perf_counter() - t0 < 10.0:
        one_layer()
        reps += 1
        if time.perf_counter() - t0 > 30.0:
            break
    t = (time.perf_counter() - t0) / reps
    cpu = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": seq / (32 * t), "unit": "tokens/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"1 of 32 decoder layers' 7 BinaryDiff projections (Llama-2-7B shapes) at seq {seq}, unpack inside the "
                      f"timed region (BASELINE.md 3, variant 1), {reps} reps x {t * 1e3:.0f} ms, extrapolated x32 layers; "
                      f"host: {cpu}, os.cpu_count()={os.cpu_count()}"}


def delta_gemm_microbench(dev, M=4096, N=4096, K=4096, iters=30):
    import bitdelta_amd as bd
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(1, M, K, device=dev, generator=g).bfloat16()
    p = torch.randint(-2 ** 31, 2 ** 31 - 1, (1, K // 32, N), device=dev, generator=g, dtype=torch.int64).to(torch.int32)
    out = torch.empty(1, M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(5):
        bd.delta_bmm(x, p, out=out, round_mode=0)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record()
        bd.delta_bmm(x, p, out=out, round_mode=0)
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    avg = sum(ts) / len(ts)
    fl = 2.0 * M * N * K
    return {"shape": [M, N, K], "flops_per_launch": fl, "avg_ms": avg, "median_ms": ts[len(ts) // 2],
            "tflops": fl / avg * 1e-9, "tflops_median": fl / ts[len(ts) // 2] * 1e-9,
            "frac_of_peak": fl / avg * 1e-9 / PEAK_BF16_TFLOPS, "peak_tflops": PEAK_BF16_TFLOPS,
            "bytes_per_launch": 2.0 * M * K + K * N / 8 + 2.0 * M * N}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="prefill", choices=["prefill", "mt-decode"],
                    help="prefill = BASELINE configs[1] (default, the bench line the driver records); mt-decode = configs[2]: "
                         "Mistral-7B base + T tenant deltas, batched decode steps (HBM-bound), reported for DESIGN.md")
    ap.add_argument("--model", default=None)
    ap.add_argument("--tenants", type=int, default=6)
    ap.add_argument("--kv-len", type=int, default=512)
    ap.add_argument("--seq", type=int, default=2048)
    ap.add_argument("--layers", type=int, default=None, help="debug only: fewer layers (result is then marked invalid)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    from bitdelta_amd import dist as bdd
    rank, world, local = bdd.init_from_env()
    assert world == args.gpus or world == 1, f"WORLD_SIZE={world} but --gpus {args.gpus}"
    assert torch.cuda.is_available(), "bench.py needs a ROCm device (there is no CPU path)"
    local = local % torch.cuda.device_count()        # one rank per GPU on a full node; wraps only in the gloo smoke test of this path
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from bitdelta_amd import _lib
    _lib.lib()                                   # fail loudly if the HIP library is missing
    from bench_model import Decoder

    timer = LaunchTimer()
    timer.install()
    decode = args.workload == "mt-decode"
    args.model = args.model or ("mistral-7b" if decode else "llama-2-7b")
    if decode:
        T = len(bdd.tenants_for_rank(args.tenants * world, rank, world))      # tenants are partitioned across ranks
        model = Decoder(args.model, dev, dtype=torch.float16, tenants=T, layers=args.layers, seed=1234 + rank)
        cache = model.new_cache(T, args.kv_len + args.steps + args.warmup + 8)
        model(torch.randint(0, model.cfg[5], (T, args.kv_len), device=dev), pos0=0, cache=cache)     # prefill the cache
        tok = torch.randint(0, model.cfg[5], (T, 1), device=dev)
        pos = [args.kv_len]

        def step():
            out = model(tok, pos0=pos[0], cache=cache)
            pos[0] += 1
            return out

        def step_fixed():           # same work at a fixed position (static shapes): what the hipGraph replays
            for c in cache:
                c[2] = args.kv_len
            return model(tok, pos0=args.kv_len, cache=cache)
    else:
        model = Decoder(args.model, dev, layers=args.layers, seed=1234 + rank)
        ids = torch.randint(0, model.cfg[5], (1, args.seq), device=dev)

        def step():
            return model(ids)

    for _ in range(args.warmup):
        step()
    timer.enabled = True
    dt = bdd.timed_region(step, args.steps, device_sync=torch.cuda.synchronize)
    timer.enabled = False
    torch.cuda.synchronize()
    n_launch, k_ms, k_flops, k_bytes = timer.summary()

    tokens = (args.tenants if decode else args.seq) * args.steps * world
    value = tokens / dt
    graph_ms = None
    if decode:
        # launch-bound loop -> hipGraph: capture one decode step (448 kernel launches + glue) and replay it
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    step_fixed()
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                step_fixed()
            for _ in range(3):
                g.replay()
            graph_s = bdd.timed_region(g.replay, args.steps, device_sync=torch.cuda.synchronize)
            graph_ms = graph_s / args.steps * 1e3
        except Exception as e:          # report, never hide
            graph_ms = f"capture failed: {type(e).__name__}: {e}"
    if rank != 0:
        return
    if decode:
        gbs = k_bytes / k_ms * 1e-6 if k_ms > 0 else 0.0
        out = {
            "metric": "multi-tenant batched decode tokens/s (BASELINE.json configs[2]: Mistral-7B base + T 1-bit deltas)",
            "value": value, "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"{args.model} base + {args.tenants} tenant deltas per GPU, decode step at kv length "
                                   f"{args.kv_len}, one token per tenant; {len(model.layers)} layers x 7 fused DiffCompressModule "
                                   "projections", "tenants_per_gpu": args.tenants, "parallelism": f"tenants partitioned over {world} rank(s), base replicated, no collective",
                       "valid": args.layers is None},
            "roofline": {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                         "traffic": None, "kernel": "bd::gemv_kernel (+ gemv_reduce_kernel) via bd_binary_linear",
                         "launches": n_launch, "kernel_ms_total": k_ms, "algorithmic_bytes_total": k_bytes,
                         "share_of_step_time": (k_ms / 1e3) / dt if dt > 0 else None},
            "eager_ms_per_step": dt / args.steps * 1e3,
            "hipgraph_ms_per_step": graph_ms,
        }
        if isinstance(graph_ms, float):        # the graph replay is the serving-relevant number: value reports it
            out["value"] = args.tenants * world / (graph_ms * 1e-3)
            out["ms_per_step"] = graph_ms
            lin_bytes = k_bytes / args.steps
            out["roofline"]["step_linear_bytes"] = lin_bytes
            out["roofline"]["whole_step_gbs_if_only_linears"] = lin_bytes / (graph_ms * 1e-3) * 1e-9
        if rank == 0:
            print(json.dumps(out), flush=True)
        return
    mb = delta_gemm_microbench(dev)
    achieved = k_flops / k_ms * 1e-9 if k_ms > 0 else 0.0
    traffic = None
    try:    # PMC counters cannot be read from inside the process: the per-launch figure comes from the committed rocprofv3 passes
        tj = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
        traffic = tj["fused_gemm"]["traffic_bytes_per_launch"]
        mb["traffic_bytes_per_launch"] = tj["delta_gemm_4096"]["traffic_bytes_per_launch"]
    except Exception:
        pass
    out = {
        "metric": "W1A16 binary-delta GEMM TFLOP/s + tokens/s, Llama-2-7B+Vicuna delta, 1/2/4/8 MI355X "
                  "(value = end-to-end prefill tokens/s; delta_gemm.tflops = the GEMM figure)",
        "value": value, "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"{args.model} base + one 1-bit delta, prefill seq {args.seq}, batch 1 per GPU "
                               f"(BASELINE.json configs[1]); {len(model.layers)} layers x 7 fused BinaryDiff projections",
                   "seq_len": args.seq, "global_batch": world, "parallelism": f"dp{world} independent replicas, no collective",
                   "valid": args.layers is None},
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / PEAK_BF16_TFLOPS, "traffic": traffic,
                     "traffic_note": "bytes per launch, (2*FETCH_SIZE + WRITE_SIZE)*1024 from separate rocprofv3 --pmc passes "
                                     "(profiles/r01_traffic.json); algorithmic bytes per launch = algorithmic_bytes_total / launches",
                     "algorithmic_bytes_total": k_bytes,
                     "kernel": "bd::delta_gemm_fx_kernel<bf16, 256x128 tile> (one-pass fused, two accumulator sets, full-tile ping-pong; x.W^T + alpha*(x.S), 4*M*N*K flop/launch)",
                     "launches": n_launch, "kernel_ms_total": k_ms, "algorithmic_flops_total": k_flops,
                     "share_of_step_time": (k_ms / 1e3) / dt if dt > 0 else None},
        "delta_gemm": mb,
        "linear_params": model.linear_param_count(),
    }
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    if rank == 0:                       # one JSON line per job (the contract); other ranks only contributed to the MAX time
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    try:
        main()
    finally:
        import torch.distributed as _dist
        if _dist.is_available() and _dist.is_initialized():
            _dist.destroy_process_group()
