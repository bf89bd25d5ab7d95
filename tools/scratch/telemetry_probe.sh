#!/bin/bash
# what clock / power telemetry can an ordinary user read on the GPU box without a GPU call?
for c in /sys/class/drm/card*; do
  d=$c/device
  [ -e $d/vendor ] || continue
  echo "== $c vendor $(cat $d/vendor 2>/dev/null) device $(cat $d/device 2>/dev/null)"
  ls $d | tr '\n' ' '; echo
  for f in pp_dpm_sclk pp_dpm_mclk pp_dpm_fclk current_link_speed gpu_busy_percent mem_busy_percent power_dpm_force_performance_level; do
    [ -r $d/$f ] && { echo "-- $f"; cat $d/$f 2>&1 | head -20; }
  done
  for h in $d/hwmon/hwmon*; do
    echo "-- $h: $(ls $h | tr '\n' ' ')"
    for f in $h/name $h/power1_average $h/power1_input $h/power1_cap $h/power1_cap_max $h/power1_cap_default $h/freq1_input $h/freq1_label $h/freq2_input $h/freq2_label $h/temp1_input $h/temp2_input $h/in0_input; do
      [ -r $f ] && echo "$(basename $f) = $(cat $f 2>&1)"
    done
  done
  [ -r $d/gpu_metrics ] && { echo "-- gpu_metrics size $(stat -c %s $d/gpu_metrics)"; head -c 256 $d/gpu_metrics | xxd | head -16; }
done
which amd-smi rocm-smi 2>&1
timeout 20 rocm-smi --showclocks --showpower 2>&1 | head -30
timeout 20 amd-smi metric --clock --power 2>&1 | head -60
