#!/bin/bash
# prefill attention A/B: the one-wave-per-SIMD kernel of round 6 (tests/native/ab/bd_attn_prefill64.h: built, correct, SLOWER) vs the shipped
# round-3 kernel; same harness, -DOLD selects the shipped one.  tools/attn_ab.sh <tag>  -> gpurun_out/<tag>/attn_ab.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/${1:-attn}; mkdir -p $O; cd tests/native/ab
H="hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -I../../../bitdelta_amd/csrc -I."
$H -o /tmp/a64_new attn64_bench.hip 2>/dev/null; $H -DOLD -o /tmp/a64_old attn64_bench.hip 2>/dev/null; $H -DBD_ATTN64_TRACE -o /tmp/a64_trace attn64_bench.hip 2>/dev/null
cd ../../..
for cfg in "2048 32 32 1 1 0" "2048 32 8 1 1 0" "1024 32 8 6 1 9" "4096 32 8 1 1 0" "2048 32 32 1 0 0" "512 32 8 2 1 5" "64 32 8 6 1 3"; do
  for b in new old; do echo "== $b $cfg"; timeout 120 /tmp/a64_$b $cfg 50 2>&1 | tail -2; done
done 2>&1 | tee $O/attn_ab.txt
echo "== phase stamps, non-causal 2048 x 32 heads" | tee -a $O/attn_ab.txt; timeout 60 /tmp/a64_trace 2048 32 32 1 0 0 20 | tail -34 | head -12 | tee -a $O/attn_ab.txt
