#!/bin/bash
# Board power / shader clock (hwmon, 10 Hz) around a command:  tools/power_run.sh <seconds to sample> <command ...>
# All cards of the node are visible in sysfs; the one the command runs on is taken to be the card whose power rises most.
secs=$1; shift
hws=$(ls -d /sys/class/drm/card*/device/hwmon/hwmon* 2>/dev/null)
( for i in $(seq 1 $((secs*10+12))); do
    for hw in $hws; do echo "$i $hw $(cat $hw/power1_input 2>/dev/null) $(cat $hw/freq1_input 2>/dev/null)"; done; sleep 0.1; done ) > /tmp/power_samples.txt &
sp=$!
sleep 0.6
"$@"
sleep 0.2; kill $sp 2>/dev/null; wait $sp 2>/dev/null
awk '{ if ($1<=5) { b[$2]+=$3; bn[$2]++ } else if ($1>12) { n[$2]++; p[$2]+=$3; f[$2]+=$4; if ($3>pm[$2]) pm[$2]=$3 } }
     END { best=""; for (h in n) { d=p[h]/n[h]-b[h]/bn[h]; if (best=="" || d>bd) { bd=d; best=h } }
           printf "  %s: idle %.0f W -> avg %.0f W (max %.0f) over %d samples, avg sclk %.0f MHz\n", best, b[best]/bn[best]/1e6, p[best]/n[best]/1e6, pm[best]/1e6, n[best], f[best]/n[best]/1e6 }' /tmp/power_samples.txt
