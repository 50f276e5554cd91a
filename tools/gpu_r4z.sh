#!/bin/bash
# round 4: tile walk order of the fused four-wave GEMM -- time and fabric traffic (FETCH_SIZE) per group_m
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r4z; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python tools/ab_tile_order.py > $O/time.txt 2>&1; cat $O/time.txt
for gm in 1 4 8; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_$gm -o p -- python tools/ab_tile_order.py --one $gm qkv > $O/pmc_$gm.log 2>&1
  f=$(find $O/pmc_$gm -name "*counter_collection.csv" | head -1)
  python3 - "$f" $gm <<'P'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    if 'delta_gemm_w4' in r.get('Kernel_Name', '') and r.get('Counter_Name') == 'FETCH_SIZE':
        acc[r['Dispatch_Id']].append(float(r['Counter_Value']))
v = [sum(x) for x in acc.values()]
v = v[5:] if len(v) > 10 else v
print(f"group_m {sys.argv[2]}: FETCH_SIZE per launch (raw counter sum over XCDs) median {sorted(v)[len(v)//2]:.0f} over {len(v)} launches")
P
  find $O/pmc_$gm -name "*.csv" -size +3M -delete
done
