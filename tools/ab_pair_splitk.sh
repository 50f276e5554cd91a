#!/bin/bash
# k-slice count of the split pair tiles (o / down of a multi-tenant request of short prompts): the rule against forced counts (BD_PAIR_SPLITK),
# per-shape launch time (tools/bench_mt_prefill.py, automatic dispatch column) and the whole 64-token request
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/${1:-pks}; mkdir -p $O
for ks in 0 3 4 7 8; do
  echo "== BD_PAIR_SPLITK=$ks"
  BD_PAIR_SPLITK=$ks python tools/bench_mt_prefill.py 6 64 2>&1 | grep -E " o | down " | sed 's/ | v16.*//'
  BD_PAIR_SPLITK=$ks python tools/bench_serving_prefill.py --lens 64 --reps 20 --modes fused 2>&1 | tail -1 | sed 's/.*kernels, //'
done 2>&1 | tee $O/pair_splitk.txt
