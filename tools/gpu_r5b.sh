#!/bin/bash
# round 5: MFMA-order energy A/B of the four-wave delta GEMM (B-stationary regions, shipped, vs X-stationary), soak mode
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r5b; mkdir -p $O
{
for M in 4096 16384; do
  for v in 2 23 2 23; do tools/soak.sh $v $M 3; done
done
} > $O/soak.txt 2>&1
cat $O/soak.txt
