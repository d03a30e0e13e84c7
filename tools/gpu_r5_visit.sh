#!/bin/bash
# Round-5 visit: gpurun --timeout 1500 -- 'bash tools/gpu_r5_visit.sh <tag> <section> ...'
#   tests:<pytest args>   ab:<file>  (lines: "<env assignments or -> <what> <n>" run through tools/gpu_exp_one.py, twice)
#   bench   prof:<workload>   py:<script>
tag=${1:-r5}; shift
export TMPDIR=/tmp PYTHONHASHSEED=0
mkdir -p gpurun_out
for sec in "$@"; do
  case $sec in
    box) bash tools/gpu_boxinfo.sh > gpurun_out/${tag}_box.txt 2>&1; tail -5 gpurun_out/${tag}_box.txt ;;
    tests:*)
      timeout 1500 python -m pytest ${sec#tests:} -m gpu -q -x > gpurun_out/${tag}_tests.log 2>&1; echo "tests rc=$?"
      tail -12 gpurun_out/${tag}_tests.log ;;
    alltests)
      timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/${tag}_pytest_gpu.log 2>&1; echo "pytest rc=$?"
      tail -8 gpurun_out/${tag}_pytest_gpu.log
      timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 ;;
    ab:*)
      f=${sec#ab:}; O=gpurun_out/${tag}_$(basename $f .txt).log; : > $O
      for rep in ${AB_REPS:-1 2}; do
        while read -r envs what n; do
          [ -z "$envs" ] && continue
          case $envs in \#*) continue ;; esac
          [ "$envs" = "-" ] && envs=""
          echo "== $envs $what $n" >> $O
          timeout 300 env $(echo $envs | tr ';' ' ') python tools/gpu_exp_one.py $what $n >> $O 2>&1 || echo "FAILED" >> $O
        done < $f
      done
      grep "==\| n=\|FAILED" $O | sed 's/residual/res/; s/, checksum.*perm/ perm/' ;;
    bench)
      timeout 1200 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"
      cat gpurun_out/${tag}_bench.json; tail -3 gpurun_out/${tag}_bench.err ;;
    benchwl:*)
      wl=${sec#benchwl:}
      timeout 600 python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu --no-extras > gpurun_out/${tag}_bench_$wl.json 2> gpurun_out/${tag}_bench_$wl.err; echo "bench $wl rc=$?"
      cat gpurun_out/${tag}_bench_$wl.json ;;
    prof:*)
      wl=${sec#prof:}
      rm -rf gpurun_out/prof_${tag}_$wl
      timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_$wl -o $wl -- python bench.py --workload $wl --steps 10 --warmup 2 --no-extras --no-cpu > gpurun_out/prof_${tag}_$wl.log 2>&1; echo "prof $wl rc=$?"
      grep -o '"ms_per_step": [0-9.]*' gpurun_out/prof_${tag}_$wl.log
      f=$(find gpurun_out/prof_${tag}_$wl -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f gpurun_out/${tag}_${wl}_kernel_stats.csv && head -14 $f | cut -c1-180
      t=$(find gpurun_out/prof_${tag}_$wl -name '*kernel_trace.csv' | head -1)
      [ -n "$t" ] && python tools/trace_timeline.py $t > gpurun_out/${tag}_${wl}_timeline.txt 2>&1 && head -60 gpurun_out/${tag}_${wl}_timeline.txt
      # keep the merged output small: drop the raw trace
      rm -rf gpurun_out/prof_${tag}_$wl ;;
    rprof:*)
      sc=${sec#rprof:}; b=$(basename $sc .py)_${PANEL_MODE:-x}
      rm -rf gpurun_out/prof_${tag}_$b
      timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_$b -o $b -- python $sc > gpurun_out/prof_${tag}_$b.log 2>&1; echo "rprof $sc rc=$?"
      tail -2 gpurun_out/prof_${tag}_$b.log
      f=$(find gpurun_out/prof_${tag}_$b -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f gpurun_out/${tag}_${b}_kernel_stats.csv && head -12 $f | cut -c1-200
      t=$(find gpurun_out/prof_${tag}_$b -name '*kernel_trace.csv' | head -1)
      [ -n "$t" ] && python tools/trace_rows.py $t "${TRACE_GREP:-node64}" | tail -${TRACE_TAIL:-24}
      rm -rf gpurun_out/prof_${tag}_$b ;;
    py:*)
      timeout 600 python ${sec#py:} > gpurun_out/${tag}_$(basename ${sec#py:} .py).log 2>&1; echo "$sec rc=$?"
      tail -40 gpurun_out/${tag}_$(basename ${sec#py:} .py).log ;;
    *) echo "unknown section $sec" ;;
  esac
done
