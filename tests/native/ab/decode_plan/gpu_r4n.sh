#!/bin/bash
# round 4, decode plan, second session: scalar-poll barrier, rotated buffers (debug), plain reads, barrier cost
set -u
OUT=gpurun_out/r4n
mkdir -p $OUT
export TMPDIR=/tmp
run() { name=$1; shift; timeout 900 python tests/native/ab/decode_plan/ab_plan.py "$@" > $OUT/$name.txt 2>&1; echo "rc=$?" >> $OUT/$name.txt; }
run full_rotate --layers 4 --iters 100 --rotate --barriers 200
run linear_inplace --layers 4 --iters 100 --linear-only
run linear_plain --layers 4 --iters 100 --linear-only --plain
run full_plain --layers 4 --iters 100 --plain
tail -n 9 $OUT/*.txt
