#!/bin/bash
# Re-run of the bench-side evidence after the prefill glue changed (steps 6+ of tools/r03_profiles.sh, plus the attention harness and the
# serving-loop prefill latencies).   tools/r03_profiles_bench.sh <outdir>
set -u
OUT=${1:-gpurun_out/r03b}
mkdir -p "$OUT"
export TMPDIR=/tmp
{ for a in "2048 32 32 1 1 0" "2048 32 8 1 1 0" "1024 32 8 6 1 9" "4096 32 8 1 1 0" "2048 32 32 1 0 0" "64 32 8 6 1 3"; do timeout 60 tests/native/attn_bench $a 50; done; } > "$OUT/prefill_attention.txt" 2>&1
timeout 300 python tools/bench_serving_prefill.py > "$OUT/serving_prefill.txt" 2>&1
timeout 120 python tools/ab_glue_prefill.py > "$OUT/prefill_glue.txt" 2>&1
timeout 600 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
timeout 900 tools/prof_bench.sh "$OUT/prof_bench" > "$OUT/prof_bench.log" 2>&1
timeout 300 python bench.py --workload mt-decode --steps 20 --warmup 3 --no-cpu-baseline > "$OUT/bench_mt_decode.json" 2> "$OUT/bench_mt_decode.err"
rm -rf "$OUT/prof_bench/trace" "$OUT/prof_bench/fetch" "$OUT/prof_bench/write"
ls -la "$OUT" "$OUT/prof_bench"
