export TMPDIR=/tmp
mkdir -p gpurun_out/pf
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pf/trace -o t -- python tools/bench_serving_prefill.py --lens 64 --reps 10 > gpurun_out/pf/log.txt 2>&1
tail -1 gpurun_out/pf/log.txt
f=$(find gpurun_out/pf/trace -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<PY
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot / 1e6)
for r in rows[:24]:
    print(r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us", round(float(r["Percentage"]), 1), "%", r["Name"][:100])
PY
find gpurun_out/pf -name "*.csv" -size +2M -delete; find gpurun_out/pf -name "*.db" -delete
