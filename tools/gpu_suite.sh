#!/bin/bash
# full GPU test suite + default bench line (what the driver runs at round end)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/${1:-suite}; mkdir -p $O
timeout 1500 python -m pytest tests/ -x -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_gpu.txt
timeout 900 python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
python3 - "$O" <<'P'
import json, sys
d = json.load(open(sys.argv[1] + '/bench.json'))
print('value', d['value'], 'ms/step', d['ms_per_step'], 'roofline', d['roofline']['frac'])
print('delta_gemm', [(r['shape'][0], round(r['frac_of_peak'], 3)) for r in d['delta_gemm']])
print('vendor_gemm', [(r['shape'][0], round(r['frac_of_peak'], 3)) for r in d['vendor_gemm']])
m = d.get('mt_decode', {}); print('mt_decode', m.get('hipgraph_ms_per_step'), m.get('linear_frac_of_hbm_peak'), m.get('step_frac_of_hbm_peak'), m.get('error'))
print('decode_7b', d.get('decode_7b'))
print('cpu', d.get('cpu_baseline', {}).get('value'), d.get('cpu_baseline', {}).get('cores'), d.get('cpu_baseline', {}).get('max_over_min_of_timed_calls'))
print('parity', json.dumps(d.get('parity'))[:1500])
P
