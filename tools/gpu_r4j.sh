#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r4j; mkdir -p $O
python tools/_chk_pair.py 2>&1 | tail -4
timeout 600 python -m pytest tests/test_gpu_serving.py tests/test_gpu_parity.py -x -q -m gpu -k "residual or binary_linear_vs_oracle or serving_loop" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.txt
timeout 600 python tools/bench_serving_prefill.py > $O/serving_prefill.txt 2>&1; tail -3 $O/serving_prefill.txt
timeout 600 python tools/bench_mt_prefill.py 6 64 32 > $O/mt_prefill.txt 2>&1
