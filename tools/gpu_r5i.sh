#!/bin/bash
# round 5: kernel trace of the decode step (8 Mistral layers, 6 tenants, hipGraph replay) with and without the RMSNorm hand-off
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/${1:-r5i}; mkdir -p $O; export TMPDIR=/tmp
for f in 0 1; do
  rocprofv3 --kernel-trace --output-format csv -d $O/trace$f -o t -- python tools/prof_handoff.py $f ${2:-6} > $O/log$f.txt 2>&1
  csv=$(find $O/trace$f -name "*kernel_trace.csv" | head -1)
  python3 - "$csv" $f <<'P' | tee $O/summary$f.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
last = rows[-1400:]                      # the last ~10 replays of the 8-layer step
per = collections.defaultdict(list); gaps = collections.defaultdict(list)
for a, b in zip(last[:-1], last[1:]):
    nm = a['Kernel_Name'][:86]
    if 'bd::' not in nm or 'binarize' in nm: continue
    per[nm].append((int(a['End_Timestamp']) - int(a['Start_Timestamp'])) / 1e3)
    gaps[nm].append((int(b['Start_Timestamp']) - int(a['End_Timestamp'])) / 1e3)
print(f"== norm_handoff = {sys.argv[2]}")
for nm, ds in sorted(per.items(), key=lambda x: -sum(x[1])):
    g = sorted(gaps[nm]); d = sorted(ds)
    print(f"  {len(ds):5d} x median {d[len(d)//2]:7.2f} us  (+ median gap after {g[len(g)//2]:5.2f})  {nm}")
P
  find $O/trace$f -name "*.csv" -delete
done
