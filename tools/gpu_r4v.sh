#!/bin/bash
# same-process A/Bs of the decode step (bench.py --workload mt-decode --ab-glue) at 6 tenants (Mistral-7B) and 1 (Llama-2-7B)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r4v; mkdir -p $O
export TMPDIR=/tmp
for T in 6 1; do
  M=mistral-7b; [ $T = 1 ] && M=llama-2-7b
  timeout 900 python bench.py --workload mt-decode --model $M --tenants $T --steps 30 --warmup 5 --ab-glue --no-cpu-baseline > $O/step_T$T.json 2> $O/step_T$T.err; echo "rc=$?"
  python - $O/step_T$T.json <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
ab = d['mt_decode']['glue_ab']
print({k: [round(x, 3) for x in v] for k, v in ab.items()}, d['mt_decode']['hipgraph_ms_per_step'])
P
done
