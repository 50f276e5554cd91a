#!/bin/bash
# round 5: four-wave pair tiles (variant 18) -- oracle parity + timing vs the 8-wave pair tiles (16 / 17)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r5l; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "test_binary_linear_vs_oracle" > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -5 $O/tests.txt
timeout 300 python tools/bench_mt_prefill.py 6 64 32 > $O/bench.txt 2>&1; cat $O/bench.txt
