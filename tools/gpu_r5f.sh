#!/bin/bash
# round 5: k loop of the four-wave kernels unrolled by four (no trickle) vs the shipped rolled loop -- cycle stamps + soak
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r5f; mkdir -p $O
timeout 200 tests/native/w4_trace 1 4096,16384 > $O/w4_trace.txt 2>&1; echo "trace rc=$?"
{ for M in 4096 16384; do for v in 2 32 2 32; do tools/soak.sh $v $M 3; done; done; for v in 11 33 11 33; do tools/soak.sh $v 16384 3; done; } > $O/soak.txt 2>&1
grep -A6 "trace" $O/w4_trace.txt | grep -v "^--" | head -90; cat $O/soak.txt
