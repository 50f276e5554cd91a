#!/bin/bash
# round 5: kernel trace of the single-delta decode step (Llama-2-7B + one delta, kv 512)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out/r5z; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python bench.py --workload mt-decode --tenants 1 --steps 20 --warmup 3 > $OUT/trace.json 2> $OUT/trace.err
S=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); cp "$S" $OUT/kernel_stats.csv
find $OUT/trace -name "*.csv" -size +2M -delete; find $OUT -name "*.db" -delete
python3 - $OUT/kernel_stats.csv <<'P'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'bd::' in r['Name'] and int(r['Calls']) > 90:
        print(f"{int(r['Calls']):6d} x avg {float(r['AverageNs'])/1e3:8.2f} us min {float(r['MinNs'])/1e3:7.2f}  {r['Name'][:100]}")
P
