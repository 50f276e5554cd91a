#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r4v; mkdir -p $O
export TMPDIR=/tmp
for T in 4 2; do
timeout 900 python bench.py --workload mt-decode --tenants $T --steps 30 --warmup 5 --ab-glue --no-cpu-baseline > $O/step_T$T.json 2> $O/step_T$T.err; echo "rc=$?"
python - $O/step_T$T.json <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
ab = d['mt_decode']['glue_ab']
print({k: [round(x, 3) for x in v] for k, v in ab.items() if 'plain' in k or 'residual_prefetch_on' in k or 'gateup' in k}, d['mt_decode']['hipgraph_ms_per_step'])
P
done
