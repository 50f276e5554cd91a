#!/bin/bash
# round 4, decode plan: poll probe + first correctness / timing A/B of the persistent decode-plan kernel
set -u
OUT=gpurun_out/r4m
mkdir -p $OUT
export TMPDIR=/tmp
(cd tests/native && timeout 120 ./poll_probe) > $OUT/poll_probe.txt 2>&1
echo "poll_probe rc=$?" >> $OUT/poll_probe.txt
timeout 900 python tests/native/ab/decode_plan/ab_plan.py --layers 4 --iters 100 --linear-only > $OUT/ab_plan_linear.txt 2>&1
echo "rc=$?" >> $OUT/ab_plan_linear.txt
timeout 900 python tests/native/ab/decode_plan/ab_plan.py --layers 4 --iters 100 > $OUT/ab_plan_full.txt 2>&1
echo "rc=$?" >> $OUT/ab_plan_full.txt
tail -n 12 $OUT/poll_probe.txt $OUT/ab_plan_linear.txt $OUT/ab_plan_full.txt
