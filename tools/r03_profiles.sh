#!/bin/bash
# Round-3 evidence run (one GPU box): every r03_* file under profiles/ comes from the outputs of this script.
#   tools/r03_profiles.sh <outdir>      (tests/native/w4_bench and w4_trace must be built: see their headers)
set -u
OUT=${1:-gpurun_out/r03}
mkdir -p "$OUT"
export TMPDIR=/tmp
T="timeout 300"
# 1. cycle budget of the four-wave kernels (s_memtime stamps of workgroup 0 / wave 0; ablations marked as such)
$T tests/native/w4_trace 5 4096,16384 > "$OUT/w4_cycles.txt" 2>&1
# 2. correctness (bit-identity with the 8-wave kernels) + event-timed launches
$T tests/native/w4_bench 30 4096,8192,16384 > "$OUT/w4_bench.txt" 2>&1
# 3. sustained launches with power / clock sampling: operand order, ablations, 8-wave kernels
{ for v in 0 2 4 7 8 9 1; do timeout 60 tools/soak.sh $v 4096 3; done
  for v in 0 2 10 11 12; do timeout 60 tools/soak.sh $v 16384 3; done; } > "$OUT/w4_energy.txt" 2>&1
# 4. PMC passes (kernel trace + counters, separate passes)
timeout 600 tools/prof_w4.sh "$OUT/prof_w4" > "$OUT/prof_w4.log" 2>&1
cp "$OUT/prof_w4/summary.txt" "$OUT/w4_pmc.txt" 2>/dev/null
# 5. the projections of a Llama-2-7B layer at prefill, separate vs fused launches
$T python tools/bench_prefill_linears.py > "$OUT/prefill_linears.txt" 2>&1
# 6. the bench line and its rocprofv3 passes
timeout 600 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
timeout 900 tools/prof_bench.sh "$OUT/prof_bench" > "$OUT/prof_bench.log" 2>&1
timeout 300 python bench.py --workload mt-decode --steps 20 --warmup 3 --no-cpu-baseline > "$OUT/bench_mt_decode.json" 2> "$OUT/bench_mt_decode.err"
rm -rf "$OUT/prof_w4"/*/pmc* "$OUT/prof_w4"/*/trace "$OUT/prof_bench/trace" "$OUT/prof_bench/fetch" "$OUT/prof_bench/write"
ls -la "$OUT"
