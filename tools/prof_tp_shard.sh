#!/bin/bash
# kernel trace of one rank's Llama-2-70B TP = 8 decode step (exchange stubbed), per-kernel medians of the graph-replayed launches
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out/${1:-tp}; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python tools/trace_tp_shard.py ${2:-8} > $O/log.txt 2>&1
tail -1 $O/log.txt
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 tools/trace_summary.py "$f" --last 600 --filter "" --width 150 2>/dev/null | head -40 | tee $O/kernels.txt
find $O -name "*.csv" -size +2M -delete; find $O -name "*.db" -delete
