#!/bin/bash
# round 5: decode attention ring depth 2 vs 4 after the wait fixes (BD_ATTN_DEPTH), same library, alternating; then a kernel trace of the step
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out/r5x; mkdir -p $OUT; export TMPDIR=/tmp
for i in 1 2; do for dpt in 2 4; do for T in 6 1; do
  BD_ATTN_DEPTH=$dpt timeout 300 python bench.py --workload mt-decode --tenants $T --steps 20 --warmup 3 > $OUT/d_${dpt}_${i}_$T.json 2> $OUT/d.err
  python3 - $OUT/d_${dpt}_${i}_$T.json $dpt $T <<'P'
import json, sys
d = json.load(open(sys.argv[1])); m = d.get('mt_decode', d)
print('depth', sys.argv[2], 'tenants', sys.argv[3], 'hipgraph ms/step', m.get('hipgraph_ms_per_step'), m.get('hipgraph_ms_per_step_repeats'))
P
done; done; done 2>&1 | tee $OUT/ab.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python bench.py --workload mt-decode --steps 20 --warmup 3 > $OUT/trace.json 2> $OUT/trace.err
S=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); head -8 "$S" | cut -c1-150; cp "$S" $OUT/kernel_stats.csv
find $OUT/trace -name "*.csv" -size +2M -delete; find $OUT -name "*.db" -delete
