#!/bin/bash
# prefill attention: key-split 8-wave kernel (prefill_attn_ks2_kernel, -DKS2) vs the round-3 kernel, same harness (tests/native/attn_bench.hip)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/${1:-ks2}; mkdir -p $O
H="hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -Ibitdelta_amd/csrc -Itests/native"
$H -o /tmp/attn_bench tests/native/attn_bench.hip 2>/dev/null; $H -DKS2 -o /tmp/attn_bench_ks2 tests/native/attn_bench.hip 2>/dev/null
for cfg in "2048 32 32 1 1 0" "2048 32 8 1 1 0" "1024 32 8 6 1 9" "4096 32 8 1 1 0" "2048 32 32 1 0 0" "512 32 8 2 1 5" "256 32 8 6 1 3" "128 32 8 6 1 3" "64 32 8 6 1 3" "2048 8 1 1 1 0" "192 8 2 1 1 0"; do
  for b in attn_bench attn_bench_ks2; do echo "== $b $cfg"; timeout 120 /tmp/$b $cfg 50 2>&1 | tail -2; done
done 2>&1 | tee $O/attn_ks2.txt
for m in 1 2 3 4; do echo "== ks2 mode $m"; timeout 60 /tmp/attn_bench_ks2 1024 32 8 2 1 7 5 $m | tail -2; done 2>&1 | tee -a $O/attn_ks2.txt
