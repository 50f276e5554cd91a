#!/bin/bash
# round 5: kernel trace of a 6-tenant 64-token prefill (what is outside the Linear launches?)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out/r5s; mkdir -p $OUT; export TMPDIR=/tmp
python tools/bench_serving_prefill.py --lens 64 --reps 5 > $OUT/plain.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python tools/bench_serving_prefill.py --lens 64 --reps 5 > $OUT/trace.log 2>&1
S=$(ls $OUT/trace/*/*kernel_stats.csv $OUT/trace/*kernel_stats.csv 2>/dev/null | head -1)
cp "$S" $OUT/kernel_stats.csv
find $OUT/trace -name "*.csv" -size +1M -delete; find $OUT -name "*.db" -delete
cat $OUT/plain.log; head -40 $OUT/kernel_stats.csv | cut -c1-220
