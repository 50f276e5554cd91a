#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r5q; mkdir -p $O
timeout 1100 python -m pytest tests/ -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.txt; grep -E "^FAILED" $O/pytest.txt | head
timeout 300 python tools/bench_serving_prefill.py --lens 64,128,256,1024 > $O/serving_prefill.txt 2>&1; grep "^mistral" $O/serving_prefill.txt
timeout 600 python bench.py --steps 10 --warmup 3 --no-mt-decode --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python3 -c "
import json; d=json.load(open('$O/bench.json')); print('value', round(d['value']), 'ms/step', round(d['ms_per_step'],3), 'roofline', round(d['roofline']['frac'],4), 'kernel ms/step', round(d['roofline']['kernel_ms_total']/d['steps'],3))"
