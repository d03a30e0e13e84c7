#!/bin/bash
# One visit to a GPU box, by sections:  gpurun --timeout 900 -- 'bash tools/gpu_visit.sh <tag> <section> [<section> ...]'
#   qrtests   the tall one-pass QR parity tests
#   qrprof    rocprofv3 kernel trace of the QR bench       alltests  the whole -m gpu suite + smoke
#   bench     the default bench line                       prof:<wl> kernel trace of `bench.py --workload <wl>`
# Outputs go to gpurun_out/<tag>_*; copy what is to be judged into profiles/.
tag=${1:-run}; shift
export TMPDIR=/tmp
mkdir -p gpurun_out
for sec in "$@"; do
  case $sec in
    qrtests)
      timeout 1200 python -m pytest tests/test_gpu_qr.py -q -k "tall or reproducible" > gpurun_out/${tag}_qrtests.log 2>&1; echo "qrtests rc=$?"
      tail -25 gpurun_out/${tag}_qrtests.log ;;
    qrall)
      timeout 1500 python -m pytest tests/test_gpu_qr.py tests/test_gpu_extras.py -q > gpurun_out/${tag}_qrall.log 2>&1; echo "qrall rc=$?"
      tail -15 gpurun_out/${tag}_qrall.log ;;
    qrprof)
      rm -rf gpurun_out/prof_${tag}_qr
      timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_qr -o qr -- python bench.py --workload qr --steps 10 --warmup 2 --no-extras --no-cpu > gpurun_out/prof_${tag}_qr.log 2>&1; echo "prof qr rc=$?"
      f=$(find gpurun_out/prof_${tag}_qr -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f gpurun_out/${tag}_qr_kernel_stats.csv && head -25 $f | cut -c1-200 ;;
    alltests)
      timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/${tag}_pytest_gpu.log 2>&1; echo "pytest rc=$?"
      tail -8 gpurun_out/${tag}_pytest_gpu.log
      timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 ;;
    bench)
      timeout 900 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"
      cat gpurun_out/${tag}_bench.json ;;
    prof:*)
      wl=${sec#prof:}
      rm -rf gpurun_out/prof_${tag}_$wl
      timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_$wl -o $wl -- python bench.py --workload $wl --steps 10 --warmup 2 --no-extras --no-cpu > gpurun_out/prof_${tag}_$wl.log 2>&1; echo "prof $wl rc=$?"
      grep -o '"ms_per_step": [0-9.]*' gpurun_out/prof_${tag}_$wl.log
      f=$(find gpurun_out/prof_${tag}_$wl -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f gpurun_out/${tag}_${wl}_kernel_stats.csv && head -14 $f | cut -c1-180 ;;
    py:*)
      timeout 600 python ${sec#py:} > gpurun_out/${tag}_$(basename ${sec#py:} .py).log 2>&1; echo "$sec rc=$?"
      tail -40 gpurun_out/${tag}_$(basename ${sec#py:} .py).log ;;
    *) echo "unknown section $sec" ;;
  esac
done
