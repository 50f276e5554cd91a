#!/bin/bash
# round 4: decode step under rocprofv3 with the final defaults (gate|up RMSNorm fused) -> kernel stats for profiles/r04_decode_step.txt
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r4t; mkdir -p $O
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/dstep -o t -- python bench.py --workload mt-decode --steps 20 --warmup 3 --no-cpu-baseline > $O/decode_step_under_rocprof.json 2> $O/dstep.err
cp $(find $O/dstep -name "*kernel_stats.csv" | head -1) $O/decode_step_kernel_stats.csv
find $O/dstep -name "*.csv" -size +2M -delete
grep "bd::" $O/decode_step_kernel_stats.csv | cut -c1-260
tail -c 600 $O/decode_step_under_rocprof.json
