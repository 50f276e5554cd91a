#!/bin/bash
# round 5: four-wave pair tiles as the automatic choice -- whole GPU suite + per-launch table + the 64-token 6-tenant request
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r5m; mkdir -p $O
timeout 1100 python -m pytest tests/ -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.txt
timeout 300 python tools/bench_mt_prefill.py 6 64 32 > $O/tiles.txt 2>&1; cat $O/tiles.txt
timeout 300 python tools/bench_serving_prefill.py > $O/serving_prefill.txt 2>&1; tail -8 $O/serving_prefill.txt
