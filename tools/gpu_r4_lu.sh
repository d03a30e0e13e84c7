#!/bin/bash
# Round-4 LU panel visit: correctness of the new panel kernel, A/B against the old one, phase accounting, kernel trace.
#   gpurun --timeout 1500 -- 'bash tools/gpu_r4_lu.sh <tag> [sections...]'   sections: tests ab phases prof dist
tag=${1:-r4lu}; shift
secs=${@:-tests ab phases prof}
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/${tag}.log
: > $O
for sec in $secs; do
  case $sec in
    tests)
      timeout 900 python -m pytest tests/test_gpu_factor.py -q -x -k "plu or lookahead_paths_fp32 or dist_lu" > gpurun_out/${tag}_tests.log 2>&1; echo "lu tests rc=$?" | tee -a $O
      tail -12 gpurun_out/${tag}_tests.log | tee -a $O ;;
    ab)
      for n in 2048 4096 8192 16384; do
        for sw in 1 2 1 2; do
          timeout 240 env FAER_HIP_LU_PANEL=$sw python tools/gpu_exp_one.py lu $n >> $O 2>&1 || echo "FAILED: panel=$sw n=$n" >> $O
        done
      done
      grep "lu n=" $O ;;
    phases)
      for n in 4096 16384; do
        timeout 240 env FAER_HIP_LIB=$PWD/faer-rs_amd/libfaer_hip_timing.so python tools/gpu_lu_phases.py $n >> gpurun_out/${tag}_phases.txt 2>&1 || echo "FAILED phases $n" >> $O
      done
      grep -v amdgpu.ids gpurun_out/${tag}_phases.txt | tail -12 ;;
    prof)
      rm -rf gpurun_out/prof_${tag}_lu
      timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_lu -o lu -- python bench.py --workload lu --steps 10 --warmup 2 --no-extras --no-cpu > gpurun_out/prof_${tag}_lu.log 2>&1; echo "prof lu rc=$?"
      grep -o '"ms_per_step": [0-9.]*' gpurun_out/prof_${tag}_lu.log
      f=$(find gpurun_out/prof_${tag}_lu -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f gpurun_out/${tag}_lu_kernel_stats.csv && head -14 $f | cut -c1-180
      t=$(find gpurun_out/prof_${tag}_lu -name '*kernel_trace.csv' | head -1); [ -n "$t" ] && python tools/trace_timeline.py $t > gpurun_out/${tag}_lu_timeline.txt 2>&1 && head -40 gpurun_out/${tag}_lu_timeline.txt
      rm -rf gpurun_out/prof_${tag}_lu ;;
    *) echo "unknown section $sec" ;;
  esac
done
