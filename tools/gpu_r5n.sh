#!/bin/bash
# round 5: what the 64-token 6-tenant prefill request spends outside its Linear launches (rocprofv3 kernel trace, per-kernel totals)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r5n; mkdir -p $O; export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python tools/bench_serving_prefill.py --lens 64 --reps 10 > $O/log.txt 2>&1
tail -1 $O/log.txt
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY' | tee $O/summary.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the HIP-path prefill calls are the LAST ones of the run: find the last 10 occurrences of the lm_head / tenant linear? simpler: take the last
# N kernels spanning ~ 10 x 13 ms = 130 ms
t_end = int(rows[-1]['End_Timestamp'])
last = [r for r in rows if int(r['Start_Timestamp']) > t_end - 125_000_000]
per = collections.defaultdict(lambda: [0, 0.0])
for r in last:
    nm = r['Kernel_Name'][:110]
    per[nm][0] += 1; per[nm][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
tot = sum(v[1] for v in per.values())
print(f"last {len(last)} kernels (~125 ms window): kernel time {tot / 1e3:.1f} ms")
for nm, (n, us) in sorted(per.items(), key=lambda x: -x[1][1])[:28]:
    print(f"  {n:6d} x {us / n:8.1f} us = {us / 1e3:7.2f} ms ({100 * us / tot:4.1f} %)  {nm}")
PY
find $O -name "*.csv" -delete; find $O -name "*.db" -delete
