#!/bin/bash
# round 5: delta_rows_kernel generalised to M > 1 / shared masks / 32-column super-tiles: parity, then timings against the streaming kernel
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out/r5v; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_properties.py -q -x -k "rows" 2>&1 | tail -40 > $OUT/prop.log; tail -3 $OUT/prop.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_shapes.py -q -x -k "delta_bmm or published" 2>&1 | tail -5 | tee $OUT/parity.log
BD_ROWS_TUNE=0 BD_ROWS_CHILD=shared timeout 300 python tools/bench_rows.py 2>&1 | grep -v amdgpu.ids | tee $OUT/shared.log
BD_ROWS_VARIANTS=-1 timeout 300 python tools/bench_rows.py 0 2>&1 | grep -v amdgpu.ids | tee $OUT/rows.log
