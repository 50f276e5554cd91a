#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r4k; mkdir -p $O
timeout 600 python bench.py --workload mt-decode --tenants 16 --steps 10 --warmup 3 --layers 8 > $O/mt16.json 2> $O/mt16.err; echo "mt16 rc=$?"; tail -2 $O/mt16.err
python3 -c "
import json; d=json.load(open('$O/mt16.json')); m=d['mt_decode']; print('T=16 (8 layers):', m['hipgraph_ms_per_step'], m['linear_frac_of_hbm_peak'], m['hipgraph_error'])"
timeout 600 python bench.py --workload tp70b --layers 2 --steps 2 --warmup 1 > $O/tp70b_w1.json 2> $O/tp.err; echo "tp rc=$?"; tail -2 $O/tp.err; cut -c1-600 $O/tp70b_w1.json
timeout 900 python bench.py --workload mt-decode --steps 20 --warmup 3 --ab-glue > $O/mt_decode_ab.json 2> $O/ab.err; echo "ab rc=$?"
python3 - <<'P'
import json
d=json.load(open('gpurun_out/r4k/mt_decode_ab.json'))['mt_decode']
print('hipgraph', d['hipgraph_ms_per_step'], 'linear frac', d['linear_frac_of_hbm_peak'], 'step frac', d['step_frac_of_hbm_peak'])
for k,v in d['glue_ab'].items(): print(k, [round(x,3) for x in v])
P
timeout 300 python -m pytest tests/test_dist_gloo.py -x -q -m gpu 2>&1 | tail -2
