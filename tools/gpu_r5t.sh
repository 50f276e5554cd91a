#!/bin/bash
# round 5: the one-mask-per-row delta kernel (variant 800): parity, then timings of its A/B arms on the published binary_bmm shapes
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out/r5t; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "test_delta_bmm_vs_oracle and (80 or None)" 2>&1 | tail -5 > $OUT/parity.log
timeout 600 python -m pytest tests/test_gpu_baseline_shapes.py -q -x -k published 2>&1 | tail -5 >> $OUT/parity.log
cat $OUT/parity.log
timeout 900 python tools/bench_rows.py 0 1 8 > $OUT/bench.log 2>&1
cat $OUT/bench.log
