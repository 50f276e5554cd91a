#!/bin/bash
# round 5: key-range splits of the decode attention launch (compile-time BD_ATTN_SPLITS: 4 shipped, 3, 2) after its wait fixes; same box, alternating
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out/r5y; mkdir -p $OUT
L=$PWD/bitdelta_amd/lib/libbitdelta_hip.so
for i in 1 2; do for tag in 4 3 2; do
  lib=$L; [ $tag != 4 ] && lib=$L.splits$tag
  for T in 6 1; do
    BD_HIP_LIB=$lib timeout 300 python bench.py --workload mt-decode --tenants $T --steps 20 --warmup 3 > $OUT/d_${tag}_${i}_$T.json 2> $OUT/d.err
    python3 - $OUT/d_${tag}_${i}_$T.json $tag $T <<'P'
import json, sys
d = json.load(open(sys.argv[1])); m = d.get('mt_decode', d)
print('splits', sys.argv[2], 'tenants', sys.argv[3], 'hipgraph ms/step', round(m.get('hipgraph_ms_per_step'), 4), [round(v, 4) for v in m.get('hipgraph_ms_per_step_repeats')])
P
  done
done; done 2>&1 | tee $OUT/ab.log
