#!/bin/bash
# rocprofv3 passes over the four-wave persistent kernels vs the 8-wave ones (tests/native/w4_bench, fixed kernel per process):
# kernel-trace durations + PMC counters in separate passes (MI355X_MICROARCH.md: never combined with sys / hip tracing).
#   tools/prof_w4.sh <outdir>
set -u
OUT=${1:-gpurun_out/prof_w4}
mkdir -p "$OUT"
export TMPDIR=/tmp
H=tests/native/w4_bench
run() {   # tag M variant
  local tag=$1 M=$2 v=$3
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$tag/trace" -o t -- $H 40 $M soakn$v > "$OUT/$tag.trace.log" 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE \
            --output-format csv -d "$OUT/$tag/pmc1" -o p -- $H 12 $M soakn$v > "$OUT/$tag.pmc1.log" 2>&1
  rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU TCC_HIT_sum TCC_MISS_sum \
            --output-format csv -d "$OUT/$tag/pmc2" -o p -- $H 12 $M soakn$v > "$OUT/$tag.pmc2.log" 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/$tag/pmc3" -o p -- $H 12 $M soakn$v > "$OUT/$tag.pmc3.log" 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/$tag/pmc4" -o p -- $H 12 $M soakn$v > "$OUT/$tag.pmc4.log" 2>&1
}
run pf_4096     4096  0
run w4lut_4096  4096  2
run w4lut_16384 16384 2
run pf_16384    16384 0
run fx_4096     4096  10
run w4f_4096    4096  11
find "$OUT" -name "*.csv" -size +4M -delete
for t in pf_4096 w4lut_4096 pf_16384 w4lut_16384 fx_4096 w4f_4096; do
  echo "==== $t"
  python3 tools/pmc_table.py $(ls $OUT/$t/trace/*/*kernel_trace.csv $OUT/$t/trace/*kernel_trace.csv 2>/dev/null | head -1) \
      $(ls $OUT/$t/pmc*/*/*counter_collection.csv $OUT/$t/pmc*/*counter_collection.csv 2>/dev/null) --match gemm
done > "$OUT/summary.txt" 2>&1
