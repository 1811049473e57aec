#!/bin/bash
# round 5, call 32: clock and power DURING the legs (hwmon sampled every 20 ms while ab_probe renders): is the pipeline power-limited or clocked down between launches?
mkdir -p gpurun_out/r05
ls /sys/class/drm/card*/device/hwmon/hwmon*/ 2>/dev/null | head -30 > gpurun_out/r05/hwmon_files.txt
for wl in "c2 --steps 6" "c3 --steps 2" "spaceship --steps 8" "c5 --steps 3"; do
  timeout 300 python tools/clock_power_probe.py $wl 2>&1 | tail -1 | tee -a gpurun_out/r05/clock_power_probe.log
done
