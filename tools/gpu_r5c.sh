#!/bin/bash
# round 5: trickled epilogue of the four-wave kernels -- bitwise check vs the 8-wave kernels, event timing, cycle stamps, soak
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/${1:-r5c}; mkdir -p $O
timeout 300 tests/native/w4_bench 30 4096,16384 > $O/w4_bench.txt 2>&1; echo "bench rc=$?"
timeout 200 tests/native/w4_trace 1 16384 > $O/w4_trace.txt 2>&1; echo "trace rc=$?"
{ for v in 2 30 2 30; do tools/soak.sh $v 16384 3; done; for v in 11 31 11 31; do tools/soak.sh $v 16384 3; done; } > $O/soak.txt 2>&1
grep -E "differing|BAD|med" $O/w4_bench.txt; grep -A9 "trace" $O/w4_trace.txt | head -70; cat $O/soak.txt
