#!/bin/bash
# round 5: the reference's published benchmark shapes through the shipped library (bench.py published_shapes block)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r5g; mkdir -p $O
python - > $O/published.txt 2>&1 <<'P'
import json, torch, bench
d = bench.published_shapes_block(torch.device("cuda", 0))
print(json.dumps(d))
for r in d["rows"]:
    print(f'{r["op"]:14s} B={r["B"]:2d} M={r["M"]:2d} N=K={r["N"]}  {r["us"]:7.2f} us ({r["timed_as"]}; eager {r["eager_us_from_python"]:.1f})  {r["tflops"]:7.2f} TF  published {r["published_tflops"]:6.3f} ({r["ratio_to_published"]:.1f}x)  masks {r["mask_gbs"]:6.0f} GB/s  {r["frac_of_hbm_peak"]:.3f} of HBM peak  variant {r["kernel_variant"]} {r["graph_error"] or ""}')
P
tail -12 $O/published.txt
