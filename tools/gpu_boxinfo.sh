#!/bin/bash
# Which population of boxes is this?  Clocks / partition modes next to LU / LLT timings, clocks sampled WHILE the LU runs.
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/boxinfo_$(date +%s).txt
{
  uname -r; grep -m1 "model name" /proc/cpuinfo
  rocm-smi --showclocks --showpower --showcomputepartition --showmemorypartition --showtemp 2>&1 | grep -i "clk\|Power (W)\|Partition:\|Temperature" | head -20
  (for i in $(seq 1 16); do rocm-smi --showclocks --showpower --showtemp 2>&1 | grep -i "sclk\|fclk\|mclk\|Power (W)\|junction" | tr '\n' ' '; echo; sleep 0.4; done) > gpurun_out/.clk.txt &
  python tools/gpu_exp_one.py lu 16384 2>&1 | grep "n="
  wait
  echo "--- clocks while the LU ran:"; cat gpurun_out/.clk.txt | sed 's/GPU\[0\]\t\t: //g' | cut -c1-260
  python tools/gpu_exp_one.py llt 16384 2>&1 | grep "n="
} > $O 2>&1
cat $O
