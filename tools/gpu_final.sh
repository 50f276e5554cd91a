#!/bin/bash
# what the driver runs at round end, in its order: smoke(), then the default bench line (timed by the wall clock around it)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/final; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.txt
t0=$(date +%s.%N)
python bench.py > $O/bench.json 2> $O/bench.err; rc=$?
t1=$(date +%s.%N)
echo "bench rc=$rc wall $(echo "$t1 - $t0" | bc) s"; tail -2 $O/bench.err
python3 - "$O" <<'P'
import json, sys
d = json.load(open(sys.argv[1] + '/bench.json'))
print('value', round(d['value']), 'ms/step', round(d['ms_per_step'], 3), 'roofline', round(d['roofline']['frac'], 4), 'ms/step with events', round(d['roofline']['ms_per_step_with_events'], 3))
print('delta_gemm', [(r['shape'][0], round(r['frac_of_peak'], 3)) for r in d['delta_gemm']])
print('vendor_gemm', [(r['shape'][0], round(r['frac_of_peak'], 3)) for r in d['vendor_gemm']])
print('mfma_ceiling', d.get('mfma_ceiling'))
m = d.get('mt_decode', {}); print('mt_decode', m.get('hipgraph_ms_per_step'), m.get('linear_frac_of_hbm_peak'), m.get('step_frac_of_hbm_peak'), m.get('error'))
print('decode_7b', d.get('decode_7b', {}).get('hipgraph_ms_per_step'))
print('published', [(r['op'], r['B'], r['M'], r['N'], round(r['us'], 1), round(r['ratio_to_published'], 1)) for r in d['published_shapes']['rows']])
print('cpu', d.get('cpu_baseline', {}).get('value'), d.get('cpu_baseline', {}).get('cores'))
print('parity keys', list(d.get('parity', {}).keys()))
P
