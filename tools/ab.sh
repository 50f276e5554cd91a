#!/bin/bash
# ONE parameterised GPU session script (replaces the per-question tools/gpu_r4*.sh / gpu_r5*.sh of earlier rounds).
#   tools/ab.sh <tag> <mode> [args...]            output: gpurun_out/<tag>/
# modes
#   tests   [pytest args]                          pytest -m gpu subset, tail of the log
#   decode  "<arms>" [tenants ...]                 same-process decode-step A/B (tools/ab_decode_step.py), default tenants 6 1
#   trace   "<arm>" [tenants] [layers]             rocprofv3 --kernel-trace of the decode step under ONE arm, per-kernel medians
#   libs    <libA.so> <libB.so> [tenants ...]      same-box A/B of two library BUILDS on the decode step (BD_HIP_LIB)
#   bench   [bench args]                           python bench.py ..., both output lines kept
#   prefill <libA.so> <libB.so>                    tools/ab_lib.sh (prefill step, fused roofline, delta-GEMM rows; A B A B)
#   final                                          what the driver runs at round end: smoke(), pytest -m gpu, the default bench line (wall-clocked)
# Earlier rounds' sessions (tools/gpu_r4*.sh, gpu_r5*.sh: one script per question) are in the git history; their results are the profiles/ files.
# Several "mode args" groups may be chained with `--`:  tools/ab.sh r6a tests -k decode -- decode "base:0 fg_off:256"
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; TAG=$1; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
run_mode() {
  local mode=$1; shift
  case $mode in
    tests)
      timeout 1500 python -m pytest tests -m gpu -q -x "$@" 2>&1 | tail -15 | tee -a $OUT/tests.log ;;
    decode)
      local arms=$1; shift; local ts=${*:-6 1}
      for T in $ts; do
        m=mistral-7b; [ "$T" = 1 ] && m=llama-2-7b
        timeout 900 python tools/ab_decode_step.py --model $m --tenants $T --arms $arms > $OUT/decode_T$T.json 2>> $OUT/decode.log
        echo "== tenants $T ($m)" | tee -a $OUT/decode.txt; tail -n $(echo $arms | wc -w) $OUT/decode.log | tee -a $OUT/decode.txt
      done ;;
    trace)
      local arm=$1; local T=${2:-6}; local NL=${3:-8}; local nm=${arm%%:*}
      rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_$nm -o t -- python tools/ab_decode_step.py --tenants $T --layers $NL --rounds 2 --arms $arm > $OUT/trace_$nm.log 2>&1
      echo "== kernel trace, arm $arm, $T tenants, $NL layers" | tee -a $OUT/trace.txt
      python tools/trace_summary.py $OUT/trace_$nm --last $((NL * 5 * 20 + 100)) | tee -a $OUT/trace.txt
      find $OUT/trace_$nm -name "*.csv" -delete; find $OUT/trace_$nm -name "*.db" -delete ;;
    libs)
      local A=$1 B=$2; shift 2; local ts=${*:-6 1}
      for i in 1 2; do for tag in A B; do
        lib=$A; [ $tag = B ] && lib=$B
        for T in $ts; do
          m=mistral-7b; [ "$T" = 1 ] && m=llama-2-7b
          BD_HIP_LIB=$PWD/$lib timeout 600 python tools/ab_decode_step.py --model $m --tenants $T --arms $tag:0 2>&1 >/dev/null | tail -1 | sed "s/^/$tag$i T=$T /" | tee -a $OUT/libs.txt
        done
      done; done ;;
    bench)
      timeout 1500 python bench.py "$@" > $OUT/bench.out 2> $OUT/bench.err; echo "bench rc $?" | tee -a $OUT/bench.txt
      tail -1 $OUT/bench.out > $OUT/bench.json; wc -c $OUT/bench.json | tee -a $OUT/bench.txt; tail -3 $OUT/bench.err ;;
    prefill)
      bash tools/ab_lib.sh "$1" "$2" $TAG 2>&1 | tee -a $OUT/prefill.txt ;;
    final)
      python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; echo "smoke rc=$?" | tee -a $OUT/final.txt; tail -1 $OUT/smoke.txt
      timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?" | tee -a $OUT/final.txt; tail -3 $OUT/pytest_gpu.txt | tee -a $OUT/final.txt
      t0=$(date +%s.%N); timeout 900 python bench.py > $OUT/bench.out 2> $OUT/bench.err; rc=$?; t1=$(date +%s.%N)
      echo "bench rc=$rc wall $(python3 -c "print(round($t1 - $t0, 1))") s" | tee -a $OUT/final.txt; tail -1 $OUT/bench.out > $OUT/bench.json; wc -c $OUT/bench.json | tee -a $OUT/final.txt ;;
    *) echo "unknown mode $mode"; return 2 ;;
  esac
}
args=()
for a in "$@"; do
  if [ "$a" = "--" ]; then run_mode "${args[@]}"; args=(); else args+=("$a"); fi
done
[ ${#args[@]} -gt 0 ] && run_mode "${args[@]}"
exit 0
