#!/bin/bash
# round 4: quad tiles (variants 18 / 19) -- oracle parity, per-launch table, serving-loop prefill of a 64-token request
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r4q; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "binary_linear_vs_oracle" > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt
tail -n 4 $O/pytest.txt
timeout 600 python tools/bench_mt_prefill.py 6 64 32 > $O/mt_prefill.txt 2>&1; echo rc=$?; cat $O/mt_prefill.txt
timeout 600 python tools/bench_serving_prefill.py > $O/serving_prefill.txt 2>&1; echo rc=$?; tail -n 12 $O/serving_prefill.txt
