#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r4f; mkdir -p $O
timeout 900 python bench.py --workload mt-decode --steps 20 --warmup 3 --ab-glue > $O/mt_decode_ab.json 2> $O/mt_decode_ab.err; echo "rc=$?"; tail -3 $O/mt_decode_ab.err
python3 - <<'P'
import json
d=json.load(open('gpurun_out/r4f/mt_decode_ab.json'))
m=d['mt_decode']
print('hipgraph ms', m['hipgraph_ms_per_step'], m['hipgraph_ms_per_step_repeats'])
for k,v in m['glue_ab'].items(): print(k, [round(x,3) for x in v])
P
