#!/bin/bash
# round-4 GPU session A: stream ceilings, loader/consumer decode kernel check + A/B, vendor GEMM calibration, new parity tests, bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4a; mkdir -p $O
export TMPDIR=/tmp
timeout 180 tests/native/stream_probe 235 46 > $O/stream_probe.txt 2>&1; echo "stream_probe rc=$?"
timeout 240 tests/native/ring_bench check quick > $O/ring_check_quick.txt 2> $O/ring_check_quick.err; echo "ring check quick rc=$?"
tail -3 $O/ring_check_quick.txt; tail -5 $O/ring_check_quick.err
timeout 500 tests/native/ring_bench check > $O/ring_check.txt 2> $O/ring_check.err; echo "ring check rc=$?"
tail -2 $O/ring_check.txt; tail -5 $O/ring_check.err
timeout 500 tests/native/ring_bench ab 60 > $O/ring_ab.txt 2> $O/ring_ab.err; echo "ring ab rc=$?"
timeout 300 python tools/vendor_gemm.py 2.0 > $O/vendor_gemm.txt 2>&1; echo "vendor rc=$?"
timeout 400 python -m pytest tests/test_gpu_baseline_shapes.py -x -q -k "fused_launches" > $O/pytest_fused.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_fused.txt
timeout 900 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
grep -h cold_us $O/ring_ab.txt | head -70
cat $O/stream_probe.txt | head -60
cat $O/vendor_gemm.txt
