#!/bin/bash
# round 5: PMC view of the k loop unrolled by four (variants 32 / 33) next to the rolled loop (2 / 11): separate rocprofv3 passes
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out/r5r; mkdir -p $OUT; export TMPDIR=/tmp
H=tests/native/w4_bench
run() {   # tag M variant
  local tag=$1 M=$2 v=$3
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$tag/trace" -o t -- $H 40 $M soakn$v > "$OUT/$tag.trace.log" 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE \
            --output-format csv -d "$OUT/$tag/pmc1" -o p -- $H 12 $M soakn$v > "$OUT/$tag.pmc1.log" 2>&1
  rocprofv3 --pmc SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d "$OUT/$tag/pmc2" -o p -- $H 12 $M soakn$v > "$OUT/$tag.pmc2.log" 2>&1
}
run w4lut_16384_rolled 16384 2
run w4lut_16384_x4     16384 32
run w4f_16384_rolled   16384 11
run w4f_16384_x4       16384 33
find "$OUT" -name "*.csv" -size +4M -delete
for t in w4lut_16384_rolled w4lut_16384_x4 w4f_16384_rolled w4f_16384_x4; do
  echo "==== $t"
  python3 tools/pmc_table.py $(ls $OUT/$t/trace/*/*kernel_trace.csv $OUT/$t/trace/*kernel_trace.csv 2>/dev/null | head -1) \
      $(ls $OUT/$t/pmc*/*/*counter_collection.csv $OUT/$t/pmc*/*counter_collection.csv 2>/dev/null) --match gemm
done > "$OUT/summary.txt" 2>&1
find "$OUT" -name "*.csv" -delete; find "$OUT" -name "*.db" -delete
cat "$OUT/summary.txt"
