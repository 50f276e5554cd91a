cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r4e; mkdir -p $O
for cfg in "6 28672 4096 9" "6 28672 4096 25" "6 28672 4096 17" "6 4096 4096 9" "6 4096 4096 25" "6 4096 14336 9" "6 4096 14336 25" "6 6144 4096 25" "6 6144 4096 9"; do
  timeout 120 tests/native/ring_trace $cfg > "$O/trace_$(echo $cfg | tr ' ' '_').txt" 2>&1
done
grep -h "^# T=" $O/trace_*.txt
