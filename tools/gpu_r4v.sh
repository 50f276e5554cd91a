#!/bin/bash
# same-process A/Bs of the decode step (bench.py --workload mt-decode --ab-glue) at 6 tenants (Mistral-7B)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r4v; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_serving.py -x -q -m gpu -k "tile_major" 2>&1 | tail -1
timeout 900 python bench.py --workload mt-decode --steps 30 --warmup 5 --ab-glue --no-cpu-baseline > $O/step_T6.json 2> $O/step_T6.err; echo "rc=$?"
python - $O/step_T6.json <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
ab = d['mt_decode']['glue_ab']
print({k: [round(x, 3) for x in v] for k, v in ab.items() if 'resid' in k}, d['mt_decode']['hipgraph_ms_per_step'])
P
