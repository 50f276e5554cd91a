#!/bin/bash
# round 5: pure-MFMA soak -- energy per flop of v_mfma_f32_32x32x16_bf16 vs 16x16x32 on random operands (+-1 B operand, zero operands as references)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r5k; mkdir -p $O
hws=$(ls -d /sys/class/drm/card*/device/hwmon/hwmon* 2>/dev/null)
for v in 0 1 2 3 0 1; do
  ( for i in $(seq 1 42); do for hw in $hws; do echo "$i $hw $(cat $hw/power1_input 2>/dev/null) $(cat $hw/freq1_input 2>/dev/null)"; done; sleep 0.1; done ) > /tmp/s.txt &
  sp=$!
  sleep 0.6
  tests/native/probes/mfma_energy_probe $v 3
  sleep 0.2; kill $sp 2>/dev/null; wait $sp 2>/dev/null
  awk '{ if ($1<=5) { b[$2]+=$3; bn[$2]++ } else if ($1>12) { n[$2]++; p[$2]+=$3; f[$2]+=$4; if ($3>pm[$2]) pm[$2]=$3 } }
     END { best=""; for (h in n) { d=p[h]/n[h]-b[h]/bn[h]; if (best=="" || d>bd) { bd=d; best=h } }
           printf "  idle %.0f W -> avg %.0f W (max %.0f) over %d samples, avg sclk %.0f MHz\n", b[best]/bn[best]/1e6, p[best]/n[best]/1e6, pm[best]/1e6, n[best], f[best]/n[best]/1e6 }' /tmp/s.txt
done > $O/mfma_energy.txt 2>&1
cat $O/mfma_energy.txt
