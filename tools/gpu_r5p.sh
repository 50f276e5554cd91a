#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r5p; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "test_binary_linear_vs_oracle" > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $O/tests.txt
timeout 300 python tools/bench_mt_prefill.py 6 128 > $O/bench.txt 2>&1; cat $O/bench.txt
