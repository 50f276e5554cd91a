#!/bin/bash
# round 5: decode attention (validity byte kept raw, RoPE inputs ahead of the K/V ring, split merge loads batched): tests, then same-box A/B of the
# 6-tenant and 1-tenant decode step against the library built before the change (BD_HIP_LIB selects the build)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out/r5w; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_serving.py tests/test_gpu_parity.py -q -x -k "decode or serving or stream or handoff or norm" 2>&1 | tail -3 | tee $OUT/tests.log
A=$PWD/bitdelta_amd/lib/libbitdelta_hip.so.prev; B=$PWD/bitdelta_amd/lib/libbitdelta_hip.so
for i in 1 2; do for tag in A B; do
  lib=$A; [ $tag = B ] && lib=$B
  for T in 6 1; do
    BD_HIP_LIB=$lib timeout 300 python bench.py --workload mt-decode --tenants $T --steps 20 --warmup 3 > $OUT/d_${tag}${i}_$T.json 2> $OUT/d_${tag}${i}_$T.err
    python3 - $OUT/d_${tag}${i}_$T.json $tag$i $T <<'P'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    m = d.get('mt_decode', d)
    print(sys.argv[2], 'tenants', sys.argv[3], 'hipgraph ms/step', m.get('hipgraph_ms_per_step'), m.get('hipgraph_ms_per_step_repeats'), 'value', d.get('value'))
except Exception as e:
    print(sys.argv[2], 'failed', e)
P
  done
done; done 2>&1 | tee $OUT/ab.log
