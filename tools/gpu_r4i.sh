#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r4i; mkdir -p $O
timeout 1500 python -m pytest tests/ -x -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.txt
timeout 600 python tools/bench_mt_prefill.py 6 64 32 > $O/mt_prefill.txt 2>&1; grep "M=  64" $O/mt_prefill.txt | cut -c1-75
timeout 600 python tools/bench_serving_prefill.py > $O/serving_prefill.txt 2>&1; tail -12 $O/serving_prefill.txt
