#!/bin/bash
# round-4 GPU session: ring kernel check, timelines, A/B vs variant 600 (with / without nt weight loads)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r4d}; mkdir -p $O
export TMPDIR=/tmp
timeout 500 tests/native/ring_bench check > $O/ring_check.txt 2> $O/ring_check.err; echo "ring check rc=$?"
tail -1 $O/ring_check.txt; tail -5 $O/ring_check.err
for cfg in "6 4096 4096 1" "6 4096 4096 9" "6 4096 4096 11" "6 28672 4096 1" "6 28672 4096 9" "6 28672 4096 11" "6 28672 4096 3" "6 4096 14336 1" "6 4096 14336 9" "1 4096 4096 1"; do
  timeout 120 tests/native/ring_trace $cfg > "$O/trace_$(echo $cfg | tr ' ' '_').txt" 2>&1; echo "trace $cfg rc=$?"
done
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
grep -h "^# T=" $O/trace_*.txt
grep -h "avg (computed" $O/trace_6_28672_4096_9.txt $O/trace_6_28672_4096_1.txt
