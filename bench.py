#!/usr/bin/env python3
"""bench.py -- headline benchmark of the 1-bit-delta Linear hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: one rank per GPU under torch.distributed.run -- either the caller
                                                            launches it that way, or bench.py re-execs itself under the launcher)

Workload of `value` (BASELINE.json configs[1]): Llama-2-7B base + one 1-bit delta (Vicuna-7B-v1.5 shapes), prefill of one
2048-token sequence per GPU, synthetic weights/activations (SURVEY.md 8d recipe).  A "step" is one full prefill forward:
32 layers x 4 fused-Linear launches (q|k|v, o + residual, gate|up, down + residual = the layer's 7 BinaryDiff projections; the hot
path, hand-written HIP) + attention / RoPE / SwiGLU (HIP glue of this library) + norm / embedding / lm_head (stock torch).
N GPUs = N independent replicas (weak scaling, no data-path collective).

Prints ONE JSON line (rank 0) with the driver contract fields plus
  roofline      the dominant kernel (fused base+delta MFMA GEMM): algorithmic FLOPs of its launches / their summed durations,
                measured live with HIP events on the launch stream inside the timed region.  `traffic` is null unless THIS run
                measured it (PMC counters cannot be read in-process; the rocprofv3 passes live under profiles/)
  delta_gemm    the W1A16 delta-GEMM alone at K = N = 4096, M in {4096, 8192, 16384} (the north star's 70 %-of-peak target), same method
  vendor_gemm   the vendor's bf16 GEMM (torch.matmul -> hipBLASLt) at the same three shapes, same process, same warm-up: the
                calibration row for "what fraction of 2.5 PF does ANY dense bf16 GEMM reach on this board at its power cap"
  mfma_ceiling  a pure-MFMA soak in the same run (no memory traffic): what the matrix cores sustain on this board on random operands and with a
                +-1 second operand -- the ceiling the delta_gemm / vendor_gemm fractions should be read against
  published_shapes  the reference's own published kernel benchmark shapes (M in {1, 16}, B in {1, 8, 16}, N = K in {4096, 8192},
                fp16) through binary_matmul / binary_bmm, TFLOP/s in the reference's convention beside the published figure + mask GB/s
  decode_7b     SURVEY.md 8(d) C2 "plus decode steps": Llama-2-7B + ONE delta, single-sequence greedy decode tokens/s (hipGraph)
  mt_decode     BASELINE.json configs[2] in the same run: Mistral-7B base + 6 tenant deltas, batched greedy decode through the
                serving loop (fused q+k+v / gate+up launches, per-tenant embedding / norms / lm_head, argmax feedback, KV cache),
                eager and as a hipGraph replay; HBM roofline of its Linear launches
  cpu_baseline  the reference's CPU-executable form of the same projections (oracle/torch_port.py) on all host cores: variant 1
                (unpack inside the timed region) and variant 2 (pre-unpacked torch.matmul), bounded sample
Other workloads: --workload mt-decode (configs[2] / [4] as the main line), --workload tp70b (configs[3], needs --gpus 8).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _pin_from_env():
    """The pinned child of cpu_baseline(): adopt the affinity mask the parent chose BEFORE torch (and its OpenMP / MKL pools) is imported."""
    cores = os.environ.get("BD_CPU_BASELINE_CORES")
    if "--cpu-baseline-child" in sys.argv and cores and hasattr(os, "sched_setaffinity"):
        try:
            os.sched_setaffinity(0, [int(c) for c in cores.split(",") if c != ""])
        except (OSError, ValueError):
            pass


_pin_from_env()

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
        import bitdelta_amd.serving_loop as sl
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
        k.binary_linear = d.binary_linear = s.binary_linear = sl.binary_linear = timed
        orig_sw = k.binary_linear_swiglu

        def timed_swiglu(x, weight, mask, alpha):
            if not timer.enabled:
                return orig_sw(x, weight, mask, alpha)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            y = orig_sw(x, weight, mask, alpha)
            e1.record()
            B, M, K = x.shape
            N = weight.shape[0]
            nbytes = 2.0 * B * M * K + 2.0 * N * K + mask.shape[0] * K * N / 8.0 + 4.0 * mask.shape[0] + 2.0 * B * M * (N // 2)
            timer.records.append((e0, e1, 4.0 * B * M * K * N, nbytes))
            return y
        k.binary_linear_swiglu = sl.binary_linear_swiglu = timed_swiglu
        orig_dec = k.binary_linear_decode

        def timed_dec(x, weight, mask, alpha, **kw):
            if not timer.enabled:
                return orig_dec(x, weight, mask, alpha, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            y = orig_dec(x, weight, mask, alpha, **kw)
            e1.record()
            B, M, K = x.shape
            N = weight.shape[0]
            nmask = B if kw.get("layout", "tile") == "packed" else mask.shape[0]      # algorithmic: one mask per tenant, no padding
            nbytes = 2.0 * B * M * K + 2.0 * N * K + nmask * K * N / 8.0 + 4.0 * nmask + 2.0 * B * M * N
            timer.records.append((e0, e1, 4.0 * B * M * K * N, nbytes))
            return y
        k.binary_linear_decode = sl.binary_linear_decode = timed_dec

    def reset(self):
        self.records = []

    def summary(self):
        ms = sum(a.elapsed_time(b) for a, b, _, _ in self.records)
        fl = sum(f for _, _, f, _ in self.records)
        by = sum(n for _, _, _, n in self.records)
        return len(self.records), ms, fl, by


def physical_cores():
    """One hardware thread per physical core (first SMT sibling of each core), from sysfs; falls back to every visible CPU."""
    firsts = set()
    try:
        allowed = os.sched_getaffinity(0)
        for c in sorted(allowed):
            with open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list") as fh:
                sib = fh.read().strip().replace("-", ",").split(",")
            first = min(int(t) for t in sib if t != "")
            firsts.add(first if first in allowed else c)
    except (OSError, AttributeError):
        return sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    return sorted(firsts)


def cpu_baseline(seq=2048, reps=5):
    """The CPU baseline runs in a FRESH process started under the affinity mask (one hardware thread per physical core): an in-process
    os.sched_setaffinity only moves the calling thread, and OpenMP / MKL pool threads torch created earlier would keep their old mask
    (ADVICE r04).  The child reports the affinity it actually observes; that is what the bench line states."""
    import subprocess
    cores = physical_cores()

    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    # (no preexec_fn: this process has torch / HIP threads, and Python documents fork-time callbacks as unsafe then -- ADVICE r05.  The child
    #  receives the core list in BD_CPU_BASELINE_CORES and pins itself as its FIRST statement, before torch is imported: _pin_from_env below.)
    env.update(OMP_NUM_THREADS=str(len(cores)), MKL_NUM_THREADS=str(len(cores)), BD_CPU_BASELINE_CORES=",".join(str(c) for c in cores))
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", "--seq", str(seq), "--steps", str(reps)],
                       capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or not lines:
        raise RuntimeError(f"cpu_baseline child failed (rc {r.returncode}): {r.stderr[-1500:]}")
    return json.loads(lines[-1])


def cpu_baseline_measure(seq=2048, reps=5):
    """Reference-equivalent CPU path (oracle/torch_port.py) on the SAME unit of work as the GPU step: one 2048-token sequence through
    a decoder layer's 7 BinaryDiff projections (Llama-2-7B shapes).  BASELINE.md 3 hygiene: the process is pinned to ONE hardware
    thread per PHYSICAL core and torch runs that many threads (SMT siblings and over-subscription made round 3's figure swing 14x
    between boxes); every distinct projection shape gets one untimed warm-up call, then the MEDIAN of `reps` timed calls.  A layer is
    4 x [4096x4096] + 2 x [11008x4096] + 1 x [4096x11008], the model 32 layers.  Variant 1 unpacks the masks inside the timed region
    (what a CPU run of the reference does); variant 2 is the pure torch.matmul baseline with pre-unpacked signs.  Attention / norms
    are excluded (GPU side: about 10 % of the step)."""
    from oracle import torch_port as tp
    # this process was STARTED under the mask (cpu_baseline): every thread pool inherits it.  Report what is observed, not what was asked.
    observed = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    cores = observed
    old_aff = None
    old_threads = torch.get_num_threads()
    torch.set_num_threads(len(cores))
    torch.manual_seed(0)
    hid, inter = 4096, 11008
    distinct = [((hid, hid), 4), ((inter, hid), 2), ((hid, inter), 1)]          # ((out, in), how many per layer)
    t1 = t2 = 0.0
    detail, spread = [], []

    def median_of(fn):
        fn()                                          # warm-up (page faults, thread pool, oneDNN primitive cache)
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        return ts[len(ts) // 2], ts[-1] / ts[0]
    try:
        for (n_out, n_in), count in distinct:
            w = (torch.randn(n_out, n_in) * 0.02).bfloat16()
            mask = torch.randint(-2 ** 31, 2 ** 31 - 1, (n_in // 32, n_out), dtype=torch.int64).to(torch.int32)
            c = torch.tensor(4e-4)
            x = torch.randn(1, seq, n_in).bfloat16()
            a, sa = median_of(lambda: tp.forward_unpack_in_loop(x, w, mask, c))
            s_ = (tp.unpack32(mask) * 2 - 1).to(torch.bfloat16)
            b, sb = median_of(lambda: tp.forward_preunpacked(x, w, s_, c))
            t1 += count * a
            t2 += count * b
            spread += [sa, sb]
            detail.append(f"[{n_out}x{n_in}] {a:.2f} s / {b:.2f} s")
            del w, mask, s_, x
    finally:
        torch.set_num_threads(old_threads)
        try:
            if old_aff is not None:
                os.sched_setaffinity(0, old_aff)
        except OSError:
            pass
    cpu = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": seq / (32 * t1), "unit": "tokens/s", "cores": len(cores), "kind": "port",
            "matmul_only": {"value": seq / (32 * t2), "unit": "tokens/s",
                            "what": "variant 2 of BASELINE.md 3: pre-unpacked signs, torch.matmul only (x@W.T + coeff*(x@S))",
                            "ms_per_layer": t2 * 1e3},
            "max_over_min_of_timed_calls": max(spread) if spread else None,
            "sample": f"same unit as the GPU step (one {seq}-token sequence, Llama-2-7B projections): per DISTINCT projection shape one "
                      f"warm-up + the median of {reps} timed calls at seq {seq} (variant 1 / variant 2: " + ", ".join(detail) +
                      f"), layer = 4+2+1 of them = {t1:.1f} s, x32 layers; value = variant 1 (unpack inside the timed region, the "
                      f"reference's CPU-executable path); host: {cpu}, os.cpu_count()={os.cpu_count()}, fresh process whose OBSERVED affinity "
                      f"is {len(cores)} hardware threads (one per physical core when the sysfs topology is readable), torch threads="
                      f"{torch.get_num_threads()}, torch {torch.__version__}"}


def parity_block(dev):
    """Part of the CPU-baseline leg (the only place bench.py touches oracle/): the timed GPU kernels re-run on the TIMED shapes and
    compared with the C oracle (oracle/bd_oracle.c, pinned to the reference's golden vectors by tests/test_oracle_golden.py) on a
    sample of output columns -- the oracle takes the sliced W rows / packed-word columns, so every sampled column is exact over all
    rows and all of k (tests/test_gpu_baseline_shapes.py runs the same check with asserts).  SURVEY.md 8(d): the parity gate is
    reported with every perf number."""
    import bitdelta_amd as bd
    from bitdelta_amd import _lib
    from oracle import bd_oracle as o

    def ulp(a, b):
        def key(t):
            i = t.view(torch.int16).int()
            return torch.where(i < 0, -(i & 0x7fff), i)
        return (key(a) - key(b)).abs()

    def cols_of(N, n=48, seed=0):
        g = torch.Generator().manual_seed(seed)
        fixed = [0, 1, 127, 128, 255, 256, N // 2, N - 257, N - 129, N - 2, N - 1]
        return torch.tensor(sorted(set([c for c in fixed if 0 <= c < N] + torch.randint(0, N, (n,), generator=g).tolist())))

    out = {"oracle": "oracle/bd_oracle.c on sampled output columns, all rows, all of k",
           "gates": "bf16 / fp16 OUTPUTS: every element <= 1 ulp of the fp32 oracle rounded to the output type, or inside the cancellation "
                    "floor 2^-22 sqrt(K) max|ref| taken PER ROW (activation row x sampled columns), >= 99 % bit-equal.  The north star's "
                    "'1e-3 relative' holds in fp32-output mode (rel-Frobenius <= 1e-5 below) and on fp16 outputs; a bf16 OUTPUT cannot "
                    "meet it by construction -- one bf16 rounding is 2^-9 = 1.95e-3 relative worst case, ~1.7e-3 rel-Frobenius against "
                    "the unrounded fp32 oracle (rel_frobenius_16bit_vs_fp32_oracle below; the reference's own bf16 path has the same "
                    "floor plus its fp16 intermediate, SURVEY.md 7b)"}
    g = torch.Generator().manual_seed(5)
    # (1) the delta GEMM row at M = 4096 (bf16, round_mode 0 as timed; fp32-output mode as the exactness check)
    M, N, K = 4096, 4096, 4096
    x = torch.randn(1, M, K, generator=g).bfloat16()
    p = torch.randint(-2 ** 31, 2 ** 31 - 1, (1, K // 32, N), generator=g, dtype=torch.int64).to(torch.int32)
    cols = cols_of(N)
    c16 = bd.delta_bmm(x.to(dev), p.to(dev), round_mode=0)
    v = _lib.lib().bd_last_gemm_variant()
    c32 = bd.delta_bmm(x.to(dev), p.to(dev), out_dtype=torch.float32, round_mode=0)
    ref32 = o.delta_bmm(x, p[:, :, cols].contiguous(), out_dtype=torch.float32, round_mode=0)
    d = ulp(c16[:, :, cols.to(dev)].cpu().contiguous(), ref32.bfloat16())
    out["delta_gemm_4096"] = {"kernel_variant": v, "sampled_columns": len(cols), "max_ulp": int(d.max()), "n_needed_floor": 0, "n_checked": int(d.numel()),
                              "bit_equal": float((d == 0).float().mean()),
                              "fp32_mode_rel_frobenius": float(((c32[:, :, cols.to(dev)].cpu().double() - ref32.double()).norm() / ref32.double().norm()))}
    # (2) the fused Linear of the timed prefill at its largest shape (2048 x 4096 -> 11008)
    M, N = 2048, 11008
    x = torch.randn(1, M, K, generator=g).bfloat16()
    w = (torch.randn(N, K, generator=g) * 0.02).bfloat16()
    p = torch.randint(-2 ** 31, 2 ** 31 - 1, (1, K // 32, N), generator=g, dtype=torch.int64).to(torch.int32)
    al = torch.tensor([[4e-4]])
    cols = cols_of(N, seed=1)
    y16 = bd.binary_linear(x.to(dev), w.to(dev), p.to(dev), al.to(dev))
    v = _lib.lib().bd_last_gemm_variant()
    y32 = bd.binary_linear(x.to(dev), w.to(dev), p.to(dev), al.to(dev), out_dtype=torch.float32)
    ref32 = o.binary_linear(x, w[cols].contiguous(), p[:, :, cols].contiguous(), al, out_dtype=torch.float32, round_mode=0)
    got = y16[:, :, cols.to(dev)].cpu().contiguous()
    d = ulp(got, ref32.bfloat16())
    floor = 2.0 ** -22 * (K ** 0.5) * ref32.abs().amax(dim=-1, keepdim=True)      # per activation row
    ok = (d <= 1) | ((got.float() - ref32.bfloat16().float()).abs() <= floor)
    out["fused_linear_2048x4096_to_11008"] = {
        "kernel_variant": v, "sampled_columns": len(cols), "max_ulp": int(d.max()), "all_within_gate": bool(ok.all()),
        "n_needed_floor": int((d > 1).sum()), "n_checked": int(d.numel()),
        "bit_equal": float((d == 0).float().mean()),
        "fp32_mode_rel_frobenius": float(((y32[:, :, cols.to(dev)].cpu().double() - ref32.double()).norm() / ref32.double().norm())),
        "rel_frobenius_16bit_vs_fp32_oracle": float(((got.double() - ref32.double()).norm() / ref32.double().norm()))}
    # (3) the launches the headline number is made of: q|k|v as ONE Linear (three scale groups) and gate|up as ONE 8-row-interleaved
    #     Linear (two scales), M = 2048, built exactly as bench_model.DecoderLayer builds them; oracle on stored rows that straddle
    #     every group / interleave / tile boundary, each with its own scale
    from bench_model import FusedSingleTenantLinear
    for name, shapes, il8 in (("fused_qkv_2048x4096_to_12288_G3", [(4096, 4096)] * 3, False),
                              ("fused_gate_up_2048x4096_to_22016_il8", [(11008, 4096)] * 2, True)):
        gd = torch.Generator(device=dev).manual_seed(77)
        lin = FusedSingleTenantLinear(shapes, dev, torch.bfloat16, gd, interleave8=il8).lin
        N = lin.weight.shape[0]
        gsz = N // lin.groups
        cs = set([0, 1, 7, 8, 15, 16, 127, 128, 255, 256, N - 257, N - 256, N - 129, N - 128, N - 17, N - 16, N - 9, N - 8, N - 1])
        for gi in range(1, lin.groups if not il8 else 4):
            b = gi * gsz if not il8 else gi * (N // 4) // 16 * 16
            cs.update([b - 2, b - 1, b, b + 1, b + 7, b + 8])
        cs.update(torch.randint(0, N, (32,), generator=torch.Generator().manual_seed(3)).tolist())
        cols = torch.tensor(sorted(c for c in cs if 0 <= c < N))
        x = torch.randn(1, 2048, K, device=dev, generator=gd).bfloat16()
        y16 = lin(x)
        v = _lib.lib().bd_last_gemm_variant()
        y32 = lin(x, out_dtype=torch.float32)
        ca = lin.column_alpha(0)[cols.to(dev)].float().cpu().reshape(1, -1)
        ref32 = o.binary_linear(x.cpu(), lin.weight[cols.to(dev)].cpu().contiguous(), lin.mask[:, :, cols.to(dev)].cpu().contiguous(), ca,
                                G=len(cols), out_dtype=torch.float32, round_mode=0)
        got = y16[:, :, cols.to(dev)].cpu().contiguous()
        d = ulp(got, ref32.bfloat16())
        floor = 2.0 ** -22 * (K ** 0.5) * ref32.abs().amax(dim=-1, keepdim=True)      # per activation row
        ok = (d <= 1) | ((got.float() - ref32.bfloat16().float()).abs() <= floor)
        out[name] = {"kernel_variant": v, "scale_groups": lin.groups, "sampled_columns": len(cols), "max_ulp": int(d.max()),
                     "all_within_gate": bool(ok.all()), "n_needed_floor": int((d > 1).sum()), "n_checked": int(d.numel()),
                     "bit_equal": float((d == 0).float().mean()),
                     "fp32_mode_rel_frobenius": float(((y32[:, :, cols.to(dev)].cpu().double() - ref32.double()).norm() / ref32.double().norm()))}
        del lin, x, y16, y32
    return out


def committed_traffic():
    """HBM/fabric-side bytes per launch of the dominant kernels, from the committed rocprofv3 PMC passes over THIS command
    (FETCH_SIZE and WRITE_SIZE in separate runs, FETCH_SIZE x 2 per MI355X_MICROARCH.md; tools/prof_bench.sh).  A bench run cannot
    read PMC counters in-process, so the line carries the committed figure and names its source."""
    path = os.path.join(ROOT, "profiles", "r06_traffic.json")
    try:
        with open(path) as fh:
            d = json.load(fh)
        d["_source"] = "profiles/r06_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over `python bench.py`, tools/prof_bench.sh)"
        return d
    except Exception:
        return None


def delta_gemm_microbench(dev, Ms=(4096, 8192, 16384), N=4096, K=4096, iters=100, warmup=100):
    """The W1A16 delta-GEMM alone (SURVEY.md 8d row C1': K = N = 4096, M in {4096, 8192, 16384}, one mask, random operands, round_mode 0).
    `warmup` untimed launches first: the chip's power management needs tens of milliseconds of this kernel before its clock
    settles (the first ~20 ms after a different workload run 10 % slow), and a serving / training process lives in the steady
    state.  Then `iters` launches, each between two HIP events recorded on the launch stream (no host sync in between)."""
    import bitdelta_amd as bd
    from bitdelta_amd import _lib
    rows = []
    for M in Ms:
        g = torch.Generator(device=dev).manual_seed(1)
        x = torch.randn(1, M, K, device=dev, generator=g).bfloat16()
        p = torch.randint(-2 ** 31, 2 ** 31 - 1, (1, K // 32, N), device=dev, generator=g, dtype=torch.int64).to(torch.int32)
        out = torch.empty(1, M, N, device=dev, dtype=torch.bfloat16)
        for _ in range(warmup):
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
        rows.append({"shape": [M, N, K], "flops_per_launch": fl, "avg_ms": avg, "median_ms": ts[len(ts) // 2], "min_ms": ts[0],
                     "tflops": fl / avg * 1e-9, "tflops_median": fl / ts[len(ts) // 2] * 1e-9,
                     "frac_of_peak": fl / avg * 1e-9 / PEAK_BF16_TFLOPS, "peak_tflops": PEAK_BF16_TFLOPS,
                     "bytes_per_launch": 2.0 * M * K + K * N / 8 + 2.0 * M * N, "launches": iters, "warmup_launches": warmup,
                     "kernel_variant": _lib.lib().bd_last_gemm_variant()})
        del x, p, out
    return rows


# the reference's only published kernel numbers (notebooks/binary_gemm_kernel_triton.ipynb; fp16, Triton 2.0 kernel,
# unnamed NVIDIA GPU), TFLOP/s with its `2*B*M*N*K / t` convention.  (kind, B, M, N = K) -> published TFLOP/s, notebook line
PUBLISHED_SHAPES = [
    ("binary_matmul", 1, 1, 4096, 0.341, "ipynb:595"), ("binary_matmul", 1, 1, 8192, 0.683, "ipynb:603"),
    ("binary_matmul", 1, 16, 4096, 5.46, "ipynb:677"), ("binary_matmul", 1, 16, 8192, 10.92, "ipynb:685"),
    ("binary_bmm", 16, 1, 4096, 2.43, "ipynb:759"), ("binary_bmm", 16, 1, 8192, 2.59, "ipynb:767"),
    ("binary_bmm", 8, 1, 4096, 1.61, "ipynb:935"), ("binary_bmm", 8, 1, 8192, 2.50, "ipynb:943"),
    ("binary_bmm", 1, 1, 4096, 0.260, "ipynb:1036"), ("binary_bmm", 1, 1, 8192, 0.533, "ipynb:1044"),
]


def published_shapes_block(dev, iters=100, warmup=20):
    """The reference's published benchmark shapes, re-measured with the SHIPPED library through the reference's Python surface
    (`binary_matmul` / `binary_bmm`, fp16, masks pre-packed as `ipynb:630`): `warmup` launches, then `iters` launches each between two
    HIP events on the launch stream, median.  TFLOP/s in the reference's own convention next to its published figure (other hardware:
    not like-for-like), plus what the shape is bound by here: the sign words are read once, so mask bytes / time against the 8 TB/s HBM
    roofline (`frac_of_hbm_peak` counts masks + activations + outputs)."""
    import bitdelta_amd as bd
    from bitdelta_amd import _lib
    rows = []
    for kind, B, M, NK, pub, src in PUBLISHED_SHAPES:
        g = torch.Generator(device=dev).manual_seed(B * 1000 + M * 10 + NK)
        masks = torch.randint(-2 ** 31, 2 ** 31 - 1, (B, NK // 32, NK), device=dev, generator=g, dtype=torch.int64).to(torch.int32)
        if kind == "binary_matmul":
            a = torch.randn(M, NK, device=dev, generator=g).half()
            b = masks[0].contiguous()
            fn = lambda: bd.binary_matmul(a, b)
        else:
            a = torch.randn(B, M, NK, device=dev, generator=g).half()
            b = masks
            fn = lambda: bd.binary_bmm(a, b)
        for _ in range(warmup):
            fn()
        evs = []
        for _ in range(iters):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize()
        ts = sorted(x.elapsed_time(y) for x, y in evs)
        eager_us = ts[len(ts) // 2] * 1e3              # one call from Python: host-bound below ~20 us (ctypes + torch.empty per call)
        # device time: `per` calls captured into ONE hipGraph, replayed; per-call time = replay time / per (includes the launch boundaries)
        per, graph_us, graph_err = 20, None, None
        try:
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                fn()
            torch.cuda.current_stream(dev).wait_stream(side)
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=side):
                for _ in range(per):
                    fn()
            for _ in range(3):
                gr.replay()
            reps = []
            for _ in range(max(iters // per, 5)):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                gr.replay()
                e1.record()
                reps.append((e0, e1))
            torch.cuda.synchronize()
            rt = sorted(x.elapsed_time(y) for x, y in reps)
            graph_us = rt[len(rt) // 2] * 1e3 / per
            del gr
        except Exception as e:                           # capture refused: the eager figure stands
            torch.cuda.synchronize()
            graph_err = f"{type(e).__name__}: {e}"
        med_us = graph_us if graph_us is not None else eager_us
        flops = 2.0 * B * M * NK * NK
        mask_bytes = B * NK * NK / 8.0
        all_bytes = mask_bytes + 2.0 * B * M * NK + 2.0 * B * M * NK
        rows.append({"op": kind, "B": B, "M": M, "N": NK, "K": NK, "dtype": "f16", "us": med_us, "eager_us_from_python": eager_us,
                     "timed_as": f"hipGraph of {per} calls" if graph_us is not None else "eager calls", "graph_error": graph_err,
                     "tflops": flops / med_us * 1e-6, "published_tflops": pub, "published_source": src,
                     "ratio_to_published": flops / med_us * 1e-6 / pub, "mask_gbs": mask_bytes / med_us * 1e-3,
                     "frac_of_hbm_peak": all_bytes / med_us * 1e-3 / PEAK_HBM_GBS,
                     "kernel_variant": _lib.lib().bd_last_gemm_variant()})
        del masks, a, b
    return {"what": "reference's published benchmark shapes (BASELINE.md section 1) through binary_matmul / binary_bmm of the shipped library; "
                    "published figures: fp16 Triton kernel on an unnamed NVIDIA GPU -- shown beside, not like-for-like",
            "method": f"{warmup} warm-up calls; `us` = device time per call inside a replayed hipGraph of 20 calls (median of replays); "
                      f"eager_us_from_python = one call between two events ({iters} calls, median; host-bound for the small shapes); "
                      "TFLOP/s = 2*B*M*N*K / t (the reference's convention)",
            "rows": rows}


def mfma_ceiling_block(secs=1.5):
    """What the matrix cores sustain on THIS board in THIS run with nothing else going on: tests/native/probes/mfma_energy_probe (pure
    v_mfma_f32_32x32x16_bf16 stream, 16 waves per CU, operands in registers, no memory traffic), `secs` seconds each on random operands and with a
    +-1 second operand (the delta GEMM's operand mix).  The calibration row next to `delta_gemm` / `vendor_gemm`: the datasheet 2.5 PF is a
    zero-operand figure; on random data the board's power cap pulls the clock down (profiles/r05_mfma_energy.txt)."""
    import re
    import subprocess
    exe = os.path.join(ROOT, "tests", "native", "probes", "mfma_energy_probe")
    if not os.path.exists(exe):
        return {"error": "tests/native/probes/mfma_energy_probe not built (python -c 'import __graft_entry__ as g; g.build()')"}
    out = {"what": "pure-MFMA soak, bf16 32x32x16, no memory traffic: the sustained rate of the matrix cores under this board's power cap",
           "seconds_each": secs}
    for key, v in (("random_operands", 0), ("pm1_second_operand", 2)):
        try:
            r = subprocess.run([exe, str(v), str(secs)], capture_output=True, text=True, timeout=60)
            m = re.search(r"-> ([0-9.]+) TF \(([0-9.]+) of", r.stdout)
            out[key] = {"tflops": float(m.group(1)), "frac_of_peak": float(m.group(2))} if m else {"error": (r.stdout + r.stderr)[-300:]}
        except Exception as e:
            out[key] = {"error": f"{type(e).__name__}: {e}"}
    return out


def vendor_gemm_microbench(dev, Ms=(4096, 8192, 16384), N=4096, K=4096, iters=100, warmup=100):
    """Calibration of the 2.5 PF denominator: the vendor's dense bf16 GEMM (torch.matmul -> hipBLASLt) at the delta-GEMM's shapes,
    [M, 4096] x [4096, 4096]^T, same process, same 100 + 100 launches, each launch between two HIP events on the launch stream.  It
    does the same MFMA work as the delta-GEMM with a B operand 16x the bytes.  If it sits at or below the delta-GEMM rows, 2.5 PF
    is not reachable by a dense bf16 GEMM at this board's power cap; if it reaches 0.70, that argument is void."""
    rows = []
    for M in Ms:
        g = torch.Generator(device=dev).manual_seed(2)
        x = torch.randn(M, K, device=dev, generator=g).bfloat16()
        w = (torch.randn(N, K, device=dev, generator=g) * 0.02).bfloat16()
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        wt = w.t()
        for _ in range(warmup):
            torch.matmul(x, wt, out=out)
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
        for a, b in ev:
            a.record()
            torch.matmul(x, wt, out=out)
            b.record()
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) for a, b in ev)
        avg = sum(ts) / len(ts)
        fl = 2.0 * M * N * K
        rows.append({"shape": [M, N, K], "what": "torch.matmul(x[M,K] bf16, W[N,K]^T) -> hipBLASLt", "avg_ms": avg,
                     "median_ms": ts[len(ts) // 2], "tflops": fl / avg * 1e-9, "tflops_median": fl / ts[len(ts) // 2] * 1e-9,
                     "frac_of_peak": fl / avg * 1e-9 / PEAK_BF16_TFLOPS, "launches": iters, "warmup_launches": warmup})
        del x, w, out
    return rows


def run_mt_decode(dev, timer, model_name, tenants, kv_len, steps, warmup, layers=None, seed=4321, ab_glue=False):
    """configs[2] / [4]: one base + `tenants` 1-bit deltas, greedy decode steps at cache length ~kv_len through the serving loop.
    Returns a dict of eager / hipGraph step times and the HBM roofline of the Linear launches."""
    from bitdelta_amd import dist as bdd
    from bitdelta_amd.serving_loop import TenantDecoder
    dec = TenantDecoder.synthetic(model_name, tenants, dev, dtype=torch.float16, seed=seed, layers=layers,
                                  max_len=kv_len + steps + warmup + 16)
    vocab = dec.cfg[5]
    g = torch.Generator().manual_seed(seed)
    prompts = [torch.randint(1, vocab, (kv_len,), generator=g).tolist() for _ in range(tenants)]
    ids, am = dec.prepare(prompts)                       # kv_len is a power of two >= 64: no padding
    cache = dec.new_cache()
    logits = dec.prefill(ids, am, cache)
    first = torch.argmax(logits, dim=-1)
    L = ids.shape[1]
    st = {"cache": cache, "tok": first[:, None].clone(), "pos": torch.tensor([L], device=dev),
          "step": torch.tensor([1], device=dev), "stop_ids": torch.full((tenants, 1), -1, dtype=torch.long, device=dev),
          "out": torch.zeros(tenants, steps * 2 + warmup * 2 + 8, dtype=torch.long, device=dev),
          "stopped": torch.zeros(tenants, dtype=torch.bool, device=dev)}
    snap = {k: v.clone() for k, v in st.items() if torch.is_tensor(v)}
    valid0 = cache["valid"].clone()

    def restore():
        for k, v in snap.items():
            st[k].copy_(v)
        cache["valid"].copy_(valid0)

    for _ in range(warmup):
        dec._decode_step(st)
    eager_s = bdd.timed_region(lambda: dec._decode_step(st), steps, device_sync=torch.cuda.synchronize)
    restore()
    # Event-timed Linear launches.  The eager loop is HOST-bound (~6 ms of Python / ctypes per 4.6 ms step): a launch that finds the queue empty
    # has the host's time between `e0.record()` and the kernel's enqueue inside its event pair, and the faster the step's kernels get, the more
    # launches find it empty (the figure fell from 0.55 to 0.49 across two changes that made the replayed step 4.4 % faster).  So the
    # timed pass enqueues every step BEHIND a calibrated GPU-side spin that outlasts the host's enqueue time: the events then see back-to-back
    # device execution, whatever the host does.
    queued, hold = False, None
    try:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(200000); torch.cuda.synchronize()
        e0.record(); torch.cuda._sleep(2000000); e1.record(); torch.cuda.synchronize()
        t_cal = e0.elapsed_time(e1)
        cyc_per_ms = 2000000 / max(t_cal, 1e-3)
        hold_ms = min(max(2.0 * eager_s / steps * 1e3, 8.0), 60.0)               # >= twice the eager step, never more than 60 ms per step
        hold = int(cyc_per_ms * hold_ms)
        queued = 0.05 < t_cal < 500.0                                            # (a calibration outside this window: keep the eager figure)
    except Exception:                       # no spin kernel in this torch build: the eager figure, labelled as such
        queued = False
    timer.reset()
    timer.enabled = True
    for _ in range(steps):
        if queued:
            torch.cuda._sleep(hold)
        dec._decode_step(st)
    torch.cuda.synchronize()
    timer.enabled = False
    n_launch, k_ms, _, k_bytes = timer.summary()
    restore()
    graph_ms, graph_err, reps_out = None, None, None
    try:
        replay = dec._graph_runner(st)
        # steady state: the first ~100 ms of replays after the eager passes run 2-3 % slower (clocks settle); warm up for that long
        for _ in range(max(warmup, 12)):
            replay()
        restore()
        # three timed regions of `steps` replays each (state restored in between); the MEDIAN is reported: the first region after
        # the capture runs ~2 % slower than the following ones on every box
        reps = []
        for _ in range(3):
            restore()
            replay()
            reps.append(bdd.timed_region(replay, steps, device_sync=torch.cuda.synchronize) / steps * 1e3)
        graph_ms = sorted(reps)[1]
        reps_out = reps
    except Exception as e:          # report, never hide
        graph_err = f"{type(e).__name__}: {e}"
    ab = None
    if ab_glue and graph_ms is not None:
        # same process, same box, alternating: the step with RMSNorm / SwiGLU folded into the Linear launches vs separate glue launches
        ab = {"norm_handoff_ms": [], "fused_glue_ms": [], "fused_gateup_only_ms": [], "swiglu_epilogue_only_ms": [], "separate_glue_ms": []}
        runners = {}
        keep = (dec.fuse_glue, dec.fuse_qkv_norm, dec.fuse_gateup_norm)
        keep_h = dec.norm_handoff
        # (the round-2..4 arms keep their meaning: RMSNorm hand-off off; "norm_handoff_ms" = the shipped round-5 step)
        for name, flag, qn, gn, ho in (("norm_handoff_ms", True, False, False, True), ("fused_glue_ms", True, True, True, False),
                                       ("fused_gateup_only_ms", True, False, True, False), ("swiglu_epilogue_only_ms", True, False, False, False),
                                       ("separate_glue_ms", False, True, True, False)):
            dec.fuse_glue, dec.fuse_qkv_norm, dec.fuse_gateup_norm, dec.norm_handoff = flag, qn, gn, ho
            restore()
            runners[name] = dec._graph_runner(st)
        dec.norm_handoff = keep_h
        for _ in range(3):
            for name, run in runners.items():
                restore()
                run()
                ab[name].append(bdd.timed_region(run, steps, device_sync=torch.cuda.synchronize) / steps * 1e3)
        dec.fuse_glue, dec.fuse_qkv_norm, dec.fuse_gateup_norm = keep
        # non-temporal policy on the tile-major base-weight loads of the streaming kernel (bd_set_stream_tuning 16 = on, 32 = off),
        # each captured as its own graph (the dispatch decision is taken at capture time)
        from bitdelta_amd import _lib as _bl
        L = _bl.lib()
        nt_runs = {}
        # ... and the residual of the o / down launches fetched at kernel start (default) vs in the epilogue (1024)
        for name, flag in (("weight_nt_on_ms", 16), ("weight_nt_off_ms", 32), ("residual_prefetch_on_ms", 0), ("residual_prefetch_off_ms", 1024),
                           ("resident_rows_everywhere_ms", 64), ("resident_rows_nowhere_ms", 128)):
            L.bd_set_stream_tuning(flag)
            restore()
            nt_runs[name] = dec._graph_runner(st)
            ab[name] = []
        # ... and the two together: resident rows on every eligible launch with gate|up's norm fused / as its own launch
        keep2 = (dec.fuse_glue, dec.fuse_qkv_norm, dec.fuse_gateup_norm)
        for name, gn in (("resident_everywhere_gateup_norm_fused_ms", True), ("resident_everywhere_gateup_norm_separate_ms", False)):
            dec.fuse_glue, dec.fuse_qkv_norm, dec.fuse_gateup_norm = True, False, gn
            L.bd_set_stream_tuning(64)
            restore()
            nt_runs[name] = dec._graph_runner(st)
            ab[name] = []
        dec.fuse_glue, dec.fuse_qkv_norm, dec.fuse_gateup_norm = keep2
        L.bd_set_stream_tuning(0)
        for _ in range(3):
            for name, run in nt_runs.items():
                restore()
                run()
                ab[name].append(bdd.timed_region(run, steps, device_sync=torch.cuda.synchronize) / steps * 1e3)
        # shipped defaults with the row-major base weight instead of its tile-major decode copy
        from bitdelta_amd.serving_loop import FusedDeltaLinear
        FusedDeltaLinear.use_tiled = False
        restore()
        run_rm = dec._graph_runner(st)
        FusedDeltaLinear.use_tiled = True
        restore()
        run_tm = dec._graph_runner(st)
        ab["row_major_w_ms"], ab["tile_major_w_ms"] = [], []
        for _ in range(3):
            for name, run in (("row_major_w_ms", run_rm), ("tile_major_w_ms", run_tm)):
                restore()
                run()
                ab[name].append(bdd.timed_region(run, steps, device_sync=torch.cuda.synchronize) / steps * 1e3)
    lin_bytes, head_bytes = dec.linear_bytes_per_step()
    best_ms = graph_ms if graph_ms is not None else eager_s / steps * 1e3
    out = {
        "workload": f"{model_name} base + {tenants} tenant deltas, greedy decode at kv length {kv_len}, one token per tenant per step; "
                    f"{len(dec.layers)} layers x 4 fused delta-Linear launches (q+k+v, o, gate+up, down) + per-tenant embedding / "
                    "norms / lm_head; argmax fed back on the device",
        "tenants": tenants, "steps": steps, "valid": layers is None,
        "eager_ms_per_step": eager_s / steps * 1e3, "hipgraph_ms_per_step": graph_ms, "hipgraph_ms_per_step_repeats": reps_out,
        "hipgraph_error": graph_err,
        "tokens_per_s": tenants / (best_ms * 1e-3),
        "delta_linear_bytes_per_step": lin_bytes, "lm_head_bytes_per_step": head_bytes,
        "delta_linear_launches": n_launch, "delta_linear_ms_total_eager": k_ms,
        # event-timed Linear launches of the eager loop: algorithmic bytes / their summed durations
        "linear_gbs": k_bytes / k_ms * 1e-6 if k_ms > 0 else None,
        "linear_frac_of_hbm_peak": (k_bytes / k_ms * 1e-6 / PEAK_HBM_GBS) if k_ms > 0 else None,
        "linear_timed": "queued behind a GPU-side spin (host gaps excluded)" if queued else "eager loop (host gaps included)",
        "linear_note": "one HIP event pair per Linear launch.  Until late in round 5 the pairs were recorded in the plain eager loop, which is host-bound: "
                       "a launch that found the queue empty had the host's enqueue time inside its pair (0.49 - 0.55 depending on the box and on how fast "
                       "the OTHER kernels of the step were).  Now every timed step is enqueued behind a calibrated GPU-side spin, so the pairs see "
                       "back-to-back device execution.  Since round 5 the Linear launches also carry the RMSNorm's work (hand-off); "
                       "`step_frac_of_hbm_peak` (graph replay, all bytes of the step) is the end-to-end number",
        "norm_handoff": bool(getattr(dec, "norm_handoff", False)),
        # the whole step against the bytes its Linears must stream (everything else counted as overhead)
        "step_gbs": (lin_bytes + head_bytes) / (best_ms * 1e-3) * 1e-9,
        "step_frac_of_hbm_peak": (lin_bytes + head_bytes) / (best_ms * 1e-3) * 1e-9 / PEAK_HBM_GBS,
        "peak_gbs": PEAK_HBM_GBS,
    }
    if ab is not None:
        out["glue_ab"] = ab
    del dec, cache, st
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="prefill", choices=["prefill", "mt-decode", "tp70b"],
                    help="prefill = BASELINE configs[1] (default, the bench line the driver records; carries mt_decode as an extra key); "
                         "mt-decode = configs[2] / [4]: Mistral-7B base + T tenant deltas per GPU, batched greedy decode; "
                         "tp70b = configs[3]: Llama-2-70B shapes, tensor parallel over all ranks, RCCL all-reduce")
    ap.add_argument("--model", default=None)
    ap.add_argument("--tenants", type=int, default=None)
    ap.add_argument("--kv-len", type=int, default=512)
    ap.add_argument("--seq", type=int, default=2048)
    ap.add_argument("--layers", type=int, default=None, help="debug only: fewer layers (result is then marked invalid)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--ab-glue", action="store_true", help="mt-decode: also time the step with separate RMSNorm / SwiGLU launches")
    ap.add_argument("--no-mt-decode", action="store_true", help="skip the configs[2] leg of the default run")
    ap.add_argument("--cpu-baseline-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_child:                  # the pinned child of cpu_baseline(): CPU only, prints its JSON object and exits
        print(json.dumps(cpu_baseline_measure(seq=args.seq, reps=args.steps)), flush=True)
        return

    # `python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one rank per GPU (the driver's command line
    # may or may not carry the launcher; either way N ranks run, or the run fails -- it never silently measures one rank).
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stdout.flush()
        os.execv(sys.executable, cmd)

    from bitdelta_amd import dist as bdd
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:          # checked BEFORE the rendezvous (a short job would hang in it)
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={os.environ.get('WORLD_SIZE', '1')} rank(s)")
    assert torch.cuda.is_available(), "bench.py needs a ROCm device (there is no CPU path)"
    rank, world, local = bdd.init_from_env()
    import torch.distributed as tdist
    n_seen = tdist.get_world_size() if tdist.is_initialized() else 1
    if n_seen != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but torch.distributed sees {n_seen} rank(s)")
    shared_gpu_ok = os.environ.get("BD_DIST_BACKEND") == "gloo"      # control-flow test of the N-rank path on a box with fewer GPUs
    if torch.cuda.device_count() < args.gpus and not shared_gpu_ok:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but only {torch.cuda.device_count()} device(s) are visible")
    local = local % torch.cuda.device_count()        # one rank per GPU on a full node; wraps only in the gloo smoke test of this path
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from bitdelta_amd import _lib
    _lib.lib()                                   # fail loudly if the HIP library is missing
    timer = LaunchTimer()
    timer.install()

    if args.workload == "tp70b":
        from bitdelta_amd.tp import bench_tp70b
        out = bench_tp70b(args, dev, rank, world, timer)
        if rank == 0:
            print(json.dumps(out), flush=True)
        return

    if args.workload == "mt-decode":
        tenants = args.tenants or 6
        model = args.model or "mistral-7b"
        d = run_mt_decode(dev, timer, model, tenants, args.kv_len, args.steps, args.warmup, layers=args.layers, seed=4321 + rank,
                          ab_glue=args.ab_glue)
        ms = d["hipgraph_ms_per_step"] or d["eager_ms_per_step"]      # already the MAX over ranks (dist.timed_region)
        if rank != 0:
            return
        out = {
            "metric": "multi-tenant batched decode tokens/s (BASELINE.json configs[2] / configs[4]: Mistral-7B base + T 1-bit deltas per GPU)",
            "value": tenants * world / (ms * 1e-3), "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": d["workload"], "tenants_per_gpu": tenants,
                       "parallelism": f"tenants partitioned over {world} rank(s) ({tenants} per GPU), base replicated, no collective",
                       "valid": args.layers is None},
            "roofline": {"bound": "hbm", "achieved": d["linear_gbs"], "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": d["linear_frac_of_hbm_peak"], "traffic": None,
                         "kernel": "bd::gemv_stream_kernel via bd_binary_linear (event-timed in the eager loop)",
                         "launches": d["delta_linear_launches"], "step_frac_of_hbm_peak": d["step_frac_of_hbm_peak"]},
            "mt_decode": d,
            **bdd.runtime_info(),
        }
        print(json.dumps(out), flush=True)
        return

    from bench_model import Decoder
    args.model = args.model or "llama-2-7b"
    model = Decoder(args.model, dev, layers=args.layers, seed=1234 + rank)
    ids = torch.randint(0, model.cfg[5], (1, args.seq), device=dev)

    def step():
        return model(ids)

    for _ in range(args.warmup):
        step()
    # `value`: the K timed steps run CLEAN (no per-launch events inside the region).  The roofline of the dominant kernel is measured
    # right after, over the same K steps again with every fused launch bracketed by a HIP event pair on the launch stream.
    timer.enabled = False
    dt = bdd.timed_region(step, args.steps, device_sync=torch.cuda.synchronize)
    timer.enabled = True
    dt_events = bdd.timed_region(step, args.steps, device_sync=torch.cuda.synchronize)
    timer.enabled = False
    torch.cuda.synchronize()
    n_launch, k_ms, k_flops, k_bytes = timer.summary()
    tokens = args.seq * args.steps * world
    value = tokens / dt
    n_layers = len(model.layers)
    lin_params = model.linear_param_count()
    model_glue = getattr(model, "glue", "RMSNorm: torch; RoPE, causal attention (bd_srv_prefill_attention), SwiGLU: HIP kernels of this library")
    del model
    torch.cuda.empty_cache()
    if rank != 0:
        return
    mb = delta_gemm_microbench(dev)
    vg = vendor_gemm_microbench(dev)
    ceil = mfma_ceiling_block()
    traffic = committed_traffic()
    achieved = k_flops / k_ms * 1e-9 if k_ms > 0 else 0.0
    detail = {"delta_gemm": mb, "vendor_gemm": vg, "mfma_ceiling": ceil, "published_shapes": published_shapes_block(dev)}

    def fr(x, nd=4):
        """numbers of the contract line: 4 decimals for O(1) values, 4 significant digits for small ones; bools / ints / strings untouched"""
        if isinstance(x, bool) or not isinstance(x, float):
            return x
        return round(x, nd) if abs(x) >= 1e-2 or x == 0.0 else float(f"{x:.4g}")

    # The contract line stays SHORT (< 6 KB: the driver's record truncates long lines and keeps only `roofline` / `cpu_baseline` / `config` of the
    # non-contract keys): every figure the north star names sits inside `roofline`; the long blocks go to a "# bench detail" line printed BEFORE it.
    roof = {"bound": "mfma", "achieved": achieved, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_BF16_TFLOPS,
            "traffic": traffic.get("fused_gemm", {}).get("traffic_bytes_per_launch") if traffic else None,
            "traffic_source": "profiles/r06_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 / WRITE_SIZE passes over this command)" if traffic else "none",
            "traffic_over_algorithmic": fr(traffic["fused_gemm"]["traffic_bytes_per_launch"] / (k_bytes / n_launch), 3)
                                        if traffic and n_launch and traffic.get("fused_gemm") else None,
            "algorithmic_bytes_per_launch": k_bytes / n_launch if n_launch else None,
            "kernel": "bd::delta_gemm_w4_kernel<bf16, 256x128, fused> (x.W^T + alpha*(x.S), 4MNK flop/launch; tail on 128x128 tiles)",
            "launches": n_launch, "kernel_ms_total": fr(k_ms, 3), "algorithmic_flops_total": k_flops,
            "measured_in": "second pass of the same K steps, one HIP event pair per fused launch",
            "ms_per_step_with_events": fr(dt_events / args.steps * 1e3, 3),
            "share_of_step_time": fr((k_ms / 1e3) / dt_events, 4) if dt_events > 0 else None,
            # the north star's own figure: the W1A16 delta-GEMM ALONE (2MNK) at K = N = 4096, fraction of 2.5 PF per M; the vendor's bf16 GEMM and the
            # pure-MFMA soak of the same run beside it
            "delta_gemm": {str(r["shape"][0]): fr(r["frac_of_peak"]) for r in mb},
            "delta_gemm_tflops": {str(r["shape"][0]): fr(r["tflops"], 1) for r in mb},
            "vendor_gemm": {str(r["shape"][0]): fr(r["frac_of_peak"]) for r in vg},
            "mfma_ceiling": {k: fr(v.get("frac_of_peak")) for k, v in ceil.items() if isinstance(v, dict) and "frac_of_peak" in v}}
    out = {
        "metric": "W1A16 binary-delta GEMM TFLOP/s + tokens/s, Llama-2-7B+Vicuna delta, 1/2/4/8 MI355X "
                  "(value = end-to-end prefill tokens/s; roofline.delta_gemm = the GEMM figure)",
        "value": value, "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"{args.model} base + one 1-bit delta, prefill seq {args.seq}, batch 1 per GPU (BASELINE.json configs[1])",
                   "launches": f"{n_layers} layers x 4 fused-Linear launches (q|k|v, o, gate|up, down = a layer's 7 BinaryDiff projections)",
                   "seq_len": args.seq, "global_batch": world, "parallelism": f"dp{world} independent replicas, no collective",
                   "glue": model_glue, "valid": args.layers is None},
        "roofline": roof,
        **bdd.runtime_info(),
        "linear_params": lin_params,
    }
    single = world == 1 and not args.no_mt_decode
    dec_keys = ("tenants", "valid", "eager_ms_per_step", "hipgraph_ms_per_step", "hipgraph_error", "tokens_per_s", "linear_frac_of_hbm_peak",
                "step_frac_of_hbm_peak", "norm_handoff")

    def decode_leg(key, model, tenants, what):
        try:
            d = run_mt_decode(dev, timer, model, tenants, args.kv_len, 20, 3, layers=args.layers)
            detail[key] = d
            out[key] = {"what": what, **{k: fr(d.get(k)) for k in dec_keys}}
            out[key + "_ms"] = fr(d["hipgraph_ms_per_step"] or d["eager_ms_per_step"])
            roof[key + "_ms"] = out[key + "_ms"]
            roof[key + "_frac_of_hbm_peak"] = fr(d["step_frac_of_hbm_peak"])
        except Exception as e:
            out[key] = {"error": f"{type(e).__name__}: {e}"}
    if single:
        decode_leg("mt_decode", "mistral-7b", args.tenants or 6, "configs[2]: Mistral-7B base + 6 tenant deltas, batched greedy decode, kv 512, hipGraph replay")
        # SURVEY.md 8(d) C2 "plus decode steps": the headline model itself, one sequence, one delta
        decode_leg("decode_7b", "llama-2-7b", 1, "Llama-2-7B base + ONE delta, single-sequence greedy decode, hipGraph replay")
        # configs[4]'s per-GPU unit: 32 tenants sharded 4 per GPU over 8 GPUs (no collective) -- one GPU's share
        decode_leg("mt_decode_t4", "mistral-7b", 4, "configs[4] per-GPU unit: Mistral-7B base + 4 tenant deltas (32 tenants over 8 GPUs, no collective)")
        # configs[3]'s per-GPU unit: ONE rank's Llama-2-70B TP = 8 shards, all 80 layers, on this GPU; the all-reduce after o / down is STUBBED OUT
        try:
            from bitdelta_amd.tp import bench_tp70b_shard
            d = bench_tp70b_shard(dev, timer, layers=args.layers)
            detail["tp70b_shard"] = d
            out["tp70b_shard"] = {k: fr(v) for k, v in d.items() if k in ("what", "exchange", "layers", "prefill_ms", "prefill_fused_frac_of_mfma_peak",
                                                                          "decode_ms", "decode_frac_of_hbm_peak", "valid")}
            roof["tp70b_shard_decode_ms"] = fr(d.get("decode_ms"))
            roof["tp70b_shard_decode_frac_of_hbm_peak"] = fr(d.get("decode_frac_of_hbm_peak"))
            roof["tp70b_shard_prefill_ms"] = fr(d.get("prefill_ms"))
            roof["tp70b_shard_prefill_frac_of_mfma_peak"] = fr(d.get("prefill_fused_frac_of_mfma_peak"))
        except Exception as e:
            out["tp70b_shard"] = {"error": f"{type(e).__name__}: {e}"}
    if world == 1 and not args.no_cpu_baseline:
        cb = cpu_baseline()
        detail["cpu_baseline_sample"] = cb.get("sample")
        cb["sample"] = (f"one {args.seq}-token sequence through a layer's 7 projections x 32 layers; per distinct shape 1 warm-up + median of 5 "
                        f"calls; unpack inside the timed region; {cb.get('cores')} threads pinned one per physical core (full text: detail line)")
        cb["matmul_only"] = {"value": fr(cb.get("matmul_only", {}).get("value"), 3), "unit": "tokens/s", "what": "pre-unpacked signs, torch.matmul only"}
        out["cpu_baseline"] = cb
        try:
            pb = parity_block(dev)
            detail["parity"] = pb
            out["parity"] = {"oracle": "oracle/bd_oracle.c, sampled columns x all rows x all k (gate text: detail line)",
                             **{k: {kk: fr(vv, 6) for kk, vv in v.items() if kk in ("kernel_variant", "max_ulp", "all_within_gate", "n_needed_floor",
                                                                                    "n_checked", "bit_equal", "fp32_mode_rel_frobenius")}
                                for k, v in pb.items() if isinstance(v, dict)}}
        except Exception as e:
            out["parity"] = {"error": f"{type(e).__name__}: {e}"}
    print("# bench detail (long blocks; the ONE contract line follows): " + json.dumps(detail), flush=True)
    line = json.dumps(out)
    if len(line) > 6000:
        print(f"bench.py: warning: contract line is {len(line)} bytes (> 6000)", file=sys.stderr)
    print(line, flush=True)


if __name__ == "__main__":
    try:
        main()
    finally:
        import torch.distributed as _dist
        if _dist.is_available() and _dist.is_initialized():
            _dist.destroy_process_group()
