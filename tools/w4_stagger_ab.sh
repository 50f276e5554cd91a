#!/bin/bash
# per-wave stagger of the four-wave GEMM's epilogue store burst, same-box soak A/B (3 s sustained launches per arm, board power and
# shader clock sampled; tools/soak.sh): variants 32 / 33 = shipped (LUT + x4 k loop), 34 / 35 = + stagger.   tools/w4_stagger_ab.sh <tag>
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/${1:-w4s}; mkdir -p $O
for rep in 1 2; do
  for M in 4096 16384; do
    for v in 32 34; do echo "== delta-only variant $v M=$M (rep $rep)"; bash tools/soak.sh $v $M 3; done
  done
  for v in 33 35; do echo "== fused 2048 rows variant $v (rep $rep)"; bash tools/soak.sh $v 2048 3; done
done 2>&1 | tee $O/w4_stagger.txt
