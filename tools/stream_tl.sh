#!/bin/bash
# timeline of one streaming decode launch on cold weights (tests/native/stream_tl.hip): o (form 0), fine-grid q|k|v (form 1), down (form 2)
# variants: name:flags (comma-separated); "t_" builds carry the s_memtime stamps, the others time launches only.  The round-6 prologue
# experiments (uniform-row copy, kernel-argument warm, early weights, 24-bit row offsets) are tests/native/ab/stream_prologue_experiments.patch:
#   (in a scratch copy of the tree) patch -p0 < tests/native/ab/stream_prologue_experiments.patch;  TL_VARS="t_base:-DBD_STREAM_TRACE t_nowarm:-DBD_STREAM_TRACE,-DBD_NO_KARG_WARM generic:-DBD_ROWS_GENERIC=1 ..."
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/${1:-stl}; mkdir -p $O
H="hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result ${TL_INC:+-I$TL_INC} -Ibitdelta_amd/csrc"     # TL_INC: a directory with an experimental bd_gemv_stream.h
VARS=${TL_VARS:-"t_base:-DBD_STREAM_TRACE base:"}
for v in $VARS; do n=${v%%:*}; f=${v#*:}; $H ${f//,/ } -o /tmp/stream_tl_$n tests/native/stream_tl.hip 2>&1 | grep " error" & done; wait
for rep in 1 2; do for v in $VARS; do n=${v%%:*}; echo "== $n (pass $rep)"
  /tmp/stream_tl_$n 4096 4096 6 0; /tmp/stream_tl_$n 6144 4096 6 1; /tmp/stream_tl_$n 28672 4096 6 0; /tmp/stream_tl_$n 4096 14336 2 0; /tmp/stream_tl_$n 4096 14336 6 2
done; done 2>&1 | tee $O/stream_tl.txt
