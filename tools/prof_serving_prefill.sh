# kernel trace of the serving loop's prefill of a multi-tenant request (default: 6 tenants, prompts padded to 64; PF_LENS=256 ...):
# per-kernel calls / average us / share, and the request latency with and without the round-6 short-prompt fusions
export TMPDIR=/tmp
O=gpurun_out/${1:-pf}; mkdir -p $O
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python tools/bench_serving_prefill.py --lens ${PF_LENS:-64} --reps 10 --modes ${PF_MODES:-fused} > $O/log.txt 2>&1
tail -1 $O/log.txt
f=$(find $O/trace -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<PY | tee $O/kernels.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot / 1e6)
for r in rows[:28]:
    print(r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us", round(float(r["Percentage"]), 1), "%", r["Name"][:110])
PY
find $O -name "*.csv" -size +2M -delete; find $O -name "*.db" -delete
python tools/bench_serving_prefill.py --lens ${PF_LENS:-64} --reps 20 2>&1 | tail -2 | tee -a $O/kernels.txt
