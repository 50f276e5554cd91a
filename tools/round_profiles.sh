#!/bin/bash
# Per-round evidence under rocprofv3 (tools/round_profiles.sh gpurun_out/r06_profiles; copy the summaries to profiles/rNN_*) (separate passes: kernel-trace / each --pmc set; never combined with sys or hip tracing):
#   1. the bench command itself            -> <out>/bench_kernel_stats.csv, bench_under_rocprof.json, traffic.json
#   2. the 6-tenant decode step (hipGraph) -> <out>/decode_step_kernel_stats.csv
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=${1:-gpurun_out/round_profiles}
mkdir -p "$OUT"
export TMPDIR=/tmp
bash tools/prof_bench.sh "$OUT/bench" > "$OUT/prof_bench.log" 2>&1
cp "$OUT/bench/bench_kernel_stats.csv" "$OUT/bench_kernel_stats.csv" 2>/dev/null
cp "$OUT/bench/bench_under_rocprof.json" "$OUT/bench_under_rocprof.json" 2>/dev/null
cp "$OUT/bench/traffic.json" "$OUT/traffic.json" 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/dstep" -o t -- python bench.py --workload mt-decode --steps 20 --warmup 3 \
    > "$OUT/decode_step_under_rocprof.json" 2> "$OUT/dstep.err"
cp $(find "$OUT/dstep" -name "*kernel_stats.csv" | head -1) "$OUT/decode_step_kernel_stats.csv" 2>/dev/null
find "$OUT" -name "*.db" -delete
find "$OUT" -name "*.csv" -size +6M -delete
rm -rf "$OUT/bench/trace" "$OUT/bench/fetch" "$OUT/bench/write" "$OUT/dstep"
ls -la "$OUT"; head -14 "$OUT/bench_kernel_stats.csv"; head -10 "$OUT/decode_step_kernel_stats.csv"; cat "$OUT/traffic.json"
