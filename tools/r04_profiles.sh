#!/bin/bash
# Round-4 evidence under rocprofv3 (separate passes: kernel-trace / each --pmc set; never combined with sys or hip tracing):
#   1. the bench command itself            -> <out>/bench_kernel_stats.csv, bench_under_rocprof.json, traffic.json
#   2. the 6-tenant decode step (hipGraph) -> <out>/decode_step_kernel_stats.csv
#   3. per-launch decode Linears, cold weights, one ring_bench process per (shape, configuration, pass):
#        v600 default policy | v600 nt weight loads (shipped) | v600 nt + resident rows | v700 loader / consumer (harness-only)
#      -> <out>/decode_launch_table.txt   (kernel-trace median / min, WAIT / BUSY / VMEM counters, FETCH_SIZE)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=${1:-gpurun_out/r04_profiles}
mkdir -p "$OUT"
export TMPDIR=/tmp
bash tools/prof_bench.sh "$OUT/bench" > "$OUT/prof_bench.log" 2>&1
cp "$OUT/bench/bench_kernel_stats.csv" "$OUT/bench_kernel_stats.csv" 2>/dev/null
cp "$OUT/bench/bench_under_rocprof.json" "$OUT/bench_under_rocprof.json" 2>/dev/null
cp "$OUT/bench/traffic.json" "$OUT/traffic.json" 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/dstep" -o t -- python bench.py --workload mt-decode --steps 20 --warmup 3 \
    > "$OUT/decode_step_under_rocprof.json" 2> "$OUT/dstep.err"
cp $(find "$OUT/dstep" -name "*kernel_stats.csv" | head -1) "$OUT/decode_step_kernel_stats.csv" 2>/dev/null
H=tests/native/ring_bench
if [ -x $H ]; then
  run() {   # tag T N K variant tune
    local tag=$1 T=$2 N=$3 K=$4 v=$5 tune=$6
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/dl/$tag/trace" -o t -- $H one $T 1 $N $K 0 1 $v $tune 20 > "$OUT/dl_$tag.log" 2>&1
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE \
              --output-format csv -d "$OUT/dl/$tag/pmc1" -o p -- $H one $T 1 $N $K 0 1 $v $tune 6 > /dev/null 2>&1
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/dl/$tag/pmc3" -o p -- $H one $T 1 $N $K 0 1 $v $tune 6 > /dev/null 2>&1
  }
  for shp in "o 4096 4096" "qkv 6144 4096" "gateup 28672 4096" "down 4096 14336"; do
    set -- $shp
    run ${1}_v600_default 6 $2 $3 600 160
    run ${1}_v600_nt      6 $2 $3 600 144
    run ${1}_v600_nt_xres 6 $2 $3 600 80
    run ${1}_v700_ring    6 $2 $3 700 9
  done
  find "$OUT/dl" -name "*.csv" -size +4M -delete
  for d in "$OUT"/dl/*/; do
    t=$(basename $d)
    echo "==== $t   ($(grep -h cold_us $OUT/dl_$t.log | head -1 | cut -c1-400))"
    python3 tools/pmc_table.py $(ls $d/trace/*/*kernel_trace.csv $d/trace/*kernel_trace.csv 2>/dev/null | head -1) \
        $(ls $d/pmc*/*/*counter_collection.csv $d/pmc*/*counter_collection.csv 2>/dev/null) --match gemv
  done > "$OUT/decode_launch_table.txt" 2>&1
fi
find "$OUT" -name "*.db" -delete
find "$OUT" -name "*.csv" -size +6M -delete
rm -rf "$OUT/bench/trace" "$OUT/bench/fetch" "$OUT/bench/write" "$OUT/dstep" "$OUT/dl"
ls -la "$OUT"; head -30 "$OUT/decode_launch_table.txt"; head -12 "$OUT/decode_step_kernel_stats.csv"; cat "$OUT/traffic.json"
