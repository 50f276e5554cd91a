#!/bin/bash
# round 4, decode plan, third session: two-level barrier (per-XCD counters), and [attention -> o] alone as a 2-phase plan per layer
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out/r4w; mkdir -p $OUT
export TMPDIR=/tmp
T=tests/native/ab/decode_plan/ab_plan.py
timeout 900 python $T --layers 4 --iters 100 --pair --barriers 200 > $OUT/full_pair_hier.txt 2>&1; echo "rc=$?" >> $OUT/full_pair_hier.txt
BD_PLAN_FLAT_BARRIER=1 timeout 900 python $T --layers 4 --iters 100 --pair --barriers 200 > $OUT/full_pair_flat.txt 2>&1; echo "rc=$?" >> $OUT/full_pair_flat.txt
timeout 900 python $T --layers 4 --iters 100 --linear-only > $OUT/linear_hier.txt 2>&1; echo "rc=$?" >> $OUT/linear_hier.txt
tail -n 12 $OUT/*.txt
