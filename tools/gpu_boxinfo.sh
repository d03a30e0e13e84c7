#!/bin/bash
# Which population of boxes is this?  Clocks / partition modes next to one LU timing.
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/boxinfo_$(date +%s).txt
{
  rocm-smi --showclocks --showperflevel --showpower --showcomputepartition --showmemorypartition --showvoltage 2>&1 | grep -v "^$" | head -60
  rocm-smi --showmaxpower --showpids 2>&1 | grep -v "^$" | head -20
  cat /sys/class/drm/card*/device/current_compute_partition /sys/class/drm/card*/device/current_memory_partition 2>/dev/null
  uname -r; nproc; grep -m1 "model name" /proc/cpuinfo
  python tools/gpu_exp_one.py lu 16384 2>&1 | grep "n="
  python tools/gpu_exp_one.py llt 16384 2>&1 | grep "n="
  rocm-smi --showclocks 2>&1 | grep -i "sclk\|mclk\|fclk\|socclk" | head
} > $O 2>&1
cat $O
