#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r4g; mkdir -p $O
timeout 500 tests/native/ring_bench check > $O/ring_check.txt 2> $O/ring_check.err; echo "ring check rc=$?"; tail -1 $O/ring_check.txt; tail -3 $O/ring_check.err
timeout 500 tests/native/ring_bench ab 60 > $O/ring_ab.txt 2> $O/ring_ab.err; echo "ring ab rc=$?"; tail -3 $O/ring_ab.err
grep -h cold_us $O/ring_ab.txt | python3 -c "
import sys, json
from collections import defaultdict
agg = defaultdict(list)
for l in sys.stdin:
    r = json.loads(l)
    agg[(r['tag'], r['variant'], r['tune'])].append(r)
for (tag, v, tune), rs in agg.items():
    print(f\"{tag:24s} v{v} tune {tune:5d} bad {max(r['bad'] for r in rs):4d} warm {min(r['warm_us'] for r in rs):7.2f} us  cold {min(r['cold_us'] for r in rs):7.2f} us {max(r['cold_gbps'] for r in rs):5.0f} GB/s\")
"
