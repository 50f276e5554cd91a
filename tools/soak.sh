#!/bin/bash
# Power / clock sampling around a sustained run of one kernel variant of tests/native/w4_bench (soak mode).
#   tools/soak.sh <variant> <M> [seconds]     variants: 0 pf (8-wave), 1 w4 VALU, 2 w4 LUT, 10 fx (8-wave fused), 11 w4 fused
# All cards of the node are visible in sysfs; the one this process runs on is taken to be the card whose power rises most.
v=$1; M=$2; secs=${3:-4}
hws=$(ls -d /sys/class/drm/card*/device/hwmon/hwmon* 2>/dev/null)
( for i in $(seq 1 $((secs*10+12))); do
    for hw in $hws; do echo "$i $hw $(cat $hw/power1_input 2>/dev/null) $(cat $hw/freq1_input 2>/dev/null)"; done; sleep 0.1; done ) > /tmp/soak_samples.txt &
sp=$!
sleep 0.6
tests/native/w4_bench $((secs*10)) $M soak$v | grep soak
sleep 0.2; kill $sp 2>/dev/null; wait $sp 2>/dev/null
awk '{ if ($1<=5) { b[$2]+=$3; bn[$2]++ } else if ($1>12) { n[$2]++; p[$2]+=$3; f[$2]+=$4; if ($3>pm[$2]) pm[$2]=$3 } }
     END { best=""; for (h in n) { d=p[h]/n[h]-b[h]/bn[h]; if (best=="" || d>bd) { bd=d; best=h } }
           printf "  %s: idle %.0f W -> avg %.0f W (max %.0f) over %d samples, avg sclk %.0f MHz\n", best, b[best]/bn[best]/1e6, p[best]/n[best]/1e6, pm[best]/1e6, n[best], f[best]/n[best]/1e6 }' /tmp/soak_samples.txt
