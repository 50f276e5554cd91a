#!/bin/bash
# rocprofv3 passes over the streaming decode kernel, one harness process per (shape, pass): kernel-trace durations + PMC counters
# (separate --pmc passes, as MI355X_MICROARCH.md prescribes; never combined with sys/hip tracing).  Usage: tools/prof_decode.sh <outdir>
set -u
OUT=${1:-gpurun_out/prof_decode}
mkdir -p "$OUT"
export TMPDIR=/tmp
H=tests/native/bd_harness
run() {   # tag T N K variant tiled
  local tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$tag/trace" -o t -- $H dec1 "$@" 40 > "$OUT/$tag.trace.log" 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE \
            --output-format csv -d "$OUT/$tag/pmc1" -o p -- $H dec1 "$@" 10 > "$OUT/$tag.pmc1.log" 2>&1
  rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM TCC_HIT_sum TCC_MISS_sum \
            --output-format csv -d "$OUT/$tag/pmc2" -o p -- $H dec1 "$@" 10 > "$OUT/$tag.pmc2.log" 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/$tag/pmc3" -o p -- $H dec1 "$@" 10 > "$OUT/$tag.pmc3.log" 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/$tag/pmc4" -o p -- $H dec1 "$@" 10 > "$OUT/$tag.pmc4.log" 2>&1
}
# last argument: sign-word layout of the streaming kernel (0 reference [K/32,N], 1 tile-major, 2 packed)
run o_4096_T6_pack      6 4096 4096 600 2
run o_4096_T6_ref       6 4096 4096 600 0
run gateup_T6_pack      6 28672 4096 600 2
run gateup_T6_ref       6 28672 4096 600 0
run down_T6_pack        6 4096 14336 600 2
run qkv_T6_pack         6 6144 4096 600 2
run o_4096_T1_pack      1 4096 4096 600 2
run o_4096_T6_r01valu   6 4096 4096 300 0
run gateup_T6_r01valu   6 28672 4096 300 0
# keep only the small CSVs
find "$OUT" -name "*.csv" -size +4M -delete
ls -R "$OUT" | head -50
