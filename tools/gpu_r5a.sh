#!/bin/bash
# round 5: operand-VALUE energy A/B of the four-wave delta GEMM (sign LUT {-1,+1} shipped vs {0,2} / {0,1} / all-zero), soak mode
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r5a; mkdir -p $O
{
for M in 4096 16384; do
  for v in 2 20 21 22 2 20; do tools/soak.sh $v $M 3; done
done
for v in 11 13 11 13; do tools/soak.sh $v 16384 3; done
} > $O/soak.txt 2>&1
cat $O/soak.txt
