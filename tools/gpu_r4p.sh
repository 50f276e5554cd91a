#!/bin/bash
# round 4: tile-major norm-fused q|k|v launch -- parity test + glue A/B on the decode step
set -u
OUT=gpurun_out/r4p
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_serving.py -x -q -m gpu -k "tile_major or serving_loop" > $OUT/pytest.txt 2>&1; echo "rc=$?" >> $OUT/pytest.txt
timeout 1200 python bench.py --workload mt-decode --steps 30 --warmup 5 --ab-glue > $OUT/mt_decode_ab.json 2> $OUT/mt_decode_ab.err; echo "rc=$?" >> $OUT/mt_decode_ab.err
tail -n 5 $OUT/pytest.txt; cat $OUT/mt_decode_ab.json | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(d.get('mt_decode', d).get('ab_glue', d), indent=0))" 2>&1 | head -60
