#!/bin/bash
# rocprofv3 passes over the bench command itself (separate passes: kernel trace, FETCH_SIZE, WRITE_SIZE -- never combined with
# sys / hip tracing).  Writes the summaries the bench line and profiles/README.md cite:
#   <out>/bench_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary of `python bench.py ...`
#   <out>/bench_under_rocprof.json the bench line printed by the profiled run
#   <out>/traffic.json             FETCH_SIZE / WRITE_SIZE per launch of the dominant kernels (-> profiles/r03_traffic.json)
set -u
OUT=${1:-gpurun_out/prof_bench}
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-mt-decode"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- $CMD > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace.err"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o p -- $CMD > /dev/null 2> "$OUT/fetch.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o p -- $CMD > /dev/null 2> "$OUT/write.err"
cp $(find "$OUT/trace" -name "*kernel_stats.csv" | head -1) "$OUT/bench_kernel_stats.csv" 2>/dev/null
python3 tools/traffic_summary.py $(find "$OUT/fetch" -name "*counter_collection.csv" | head -1) $(find "$OUT/write" -name "*counter_collection.csv" | head -1) --bench > "$OUT/traffic.json"
find "$OUT" -name "*.csv" -size +6M -delete
find "$OUT" -name "*.db" -delete
ls -la "$OUT"
