#!/bin/bash
# Same-box A/B of two BUILDS of the library (compile-time kernel switches): bench.py's prefill step, fused-kernel roofline and the delta-GEMM
# rows, alternating A B A B.   tools/ab_lib.sh <libA.so> <libB.so> [outdir]       (BD_HIP_LIB selects the build, bitdelta_amd/_lib.py)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; A=$1; B=$2; O=gpurun_out/${3:-ab_lib}; mkdir -p $O
for i in 1 2; do
  for tag in A B; do
    lib=$A; [ $tag = B ] && lib=$B
    BD_HIP_LIB=$PWD/$lib timeout 600 python bench.py --steps 10 --warmup 3 --no-mt-decode --no-cpu-baseline > $O/bench_${tag}${i}.json 2> $O/bench_${tag}${i}.err || tail -3 $O/bench_${tag}${i}.err
    python3 - $O/bench_${tag}${i}.json $tag$i $lib <<'P'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[2], sys.argv[3], 'tokens/s %.0f  ms/step %.3f  fused frac %.4f  fused kernel ms/step %.3f |' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms_total'] / d['steps']),
      'delta_gemm', [(r['shape'][0], round(r['avg_ms'] * 1e3, 2), round(r['frac_of_peak'], 4)) for r in d['delta_gemm']])
P
  done
done
