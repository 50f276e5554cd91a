#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs) into profiles/r01_traffic.json.

usage: traffic_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <kernel-substring> [<label>]
Bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (MI355X_MICROARCH.md: FETCH_SIZE under-reports wide coalesced streams
by 2x on gfx950; both counters are in KiB and sit on the L2's fabric side, so Infinity-Cache hits are included).
"""
import csv
import json
import sys


def per_launch(path, counter, needle):
    by_dispatch = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter or needle not in r["Kernel_Name"]:
            continue
        by_dispatch[r["Dispatch_Id"]] = by_dispatch.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    vals = list(by_dispatch.values())
    return (sum(vals) / len(vals), len(vals)) if vals else (0.0, 0)


def bench_summary(fetch_csv, write_csv):
    """--bench: the two kernels the bench line quotes, with their algorithmic bytes"""
    out = {"note": "HBM/fabric-side bytes per launch from rocprofv3 PMC passes over `python bench.py --steps 3 --warmup 2 --no-cpu-baseline "
                   "--no-mt-decode` (FETCH_SIZE and WRITE_SIZE in SEPARATE runs; tools/prof_bench.sh).  bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024: "
                   "FETCH_SIZE reports half of the bytes of wide coalesced streams on gfx950 (MI355X_MICROARCH.md, HBM section).  The counters "
                   "sit on the L2's fabric side, so Infinity-Cache hits are included: an upper bound on DRAM traffic."}
    for key, needle, label, alg in (
            ("fused_gemm", "delta_gemm_w4_kernel<bd::W4Cfg<1, 256, 128, true", "bd::delta_gemm_w4_kernel<bf16,256x128,fused> (average over the launch mix of a Llama-2-7B layer at M = 2048)", None),
            ("delta_gemm_rows", "delta_gemm_w4_kernel<bd::W4Cfg<1, 256, 256, false", "bd::delta_gemm_w4_kernel<bf16,256x256,delta-only,LUT> (average over the M = 4096 / 8192 / 16384 rows)", None)):
        f, nf = per_launch(fetch_csv, "FETCH_SIZE", needle)
        w, nw = per_launch(write_csv, "WRITE_SIZE", needle)
        out[key] = {"kernel": label, "launches_fetch_pass": nf, "launches_write_pass": nw, "fetch_size_kb_avg": f, "write_size_kb_avg": w,
                    "traffic_bytes_per_launch": (2 * f + w) * 1024}
    return out


if __name__ == "__main__":
    if "--bench" in sys.argv:
        a = [x for x in sys.argv[1:] if x != "--bench"]
        print(json.dumps(bench_summary(a[0], a[1]), indent=1))
        sys.exit(0)
    fetch_csv, write_csv, needle = sys.argv[1:4]
    f, nf = per_launch(fetch_csv, "FETCH_SIZE", needle)
    w, nw = per_launch(write_csv, "WRITE_SIZE", needle)
    print(json.dumps({"kernel_substring": needle, "launches_fetch_pass": nf, "launches_write_pass": nw,
                      "fetch_size_kb_avg": f, "write_size_kb_avg": w,
                      "traffic_bytes_per_launch": (2 * f + w) * 1024}, indent=1))
