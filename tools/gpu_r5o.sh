#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r5o; mkdir -p $O
timeout 300 python tools/bench_serving_prefill.py --lens 64,128,256 > $O/serving_prefill.txt 2>&1; grep "^mistral" $O/serving_prefill.txt
timeout 900 python -m pytest tests/test_gpu_serving.py -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.txt
