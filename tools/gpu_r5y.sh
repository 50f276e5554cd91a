#!/bin/bash
# round 5: key-range splits of the decode attention launch chosen by (tenants x kv heads) (BD_ATTN_SPLITS_MAX=4 = the fixed 4 of rounds 3 - 5); same box, alternating
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out/r5y; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_serving.py -q -x 2>&1 | tail -2 | tee $OUT/tests.log
for i in 1 2; do for mx in 4 16; do
  for T in 1 2 4 6; do
    BD_ATTN_SPLITS_MAX=$mx timeout 300 python bench.py --workload mt-decode --tenants $T --steps 20 --warmup 3 > $OUT/d_${mx}_${i}_$T.json 2> $OUT/d.err
    python3 - $OUT/d_${mx}_${i}_$T.json $mx $T <<'P'
import json, sys
d = json.load(open(sys.argv[1])); m = d.get('mt_decode', d)
print('max splits', sys.argv[2], 'tenants', sys.argv[3], 'hipgraph ms/step', round(m.get('hipgraph_ms_per_step'), 4), [round(v, 4) for v in m.get('hipgraph_ms_per_step_repeats')])
P
  done
done; done 2>&1 | tee $OUT/ab.log
