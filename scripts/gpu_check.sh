#!/bin/bash
# One GPU-box round: smoke, GPU parity tests, a short bench, rocprofv3 kernel stats.  Everything lands in gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9" | head -4 > gpurun_out/device.txt
nproc >> gpurun_out/device.txt; lscpu | grep -E "Model name|Socket|NUMA node\(s\)" >> gpurun_out/device.txt
timeout 600 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" | tee -a gpurun_out/summary.txt
timeout 1500 python -m pytest tests -m gpu -q --timeout=600 ${PYTEST_ARGS} > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/summary.txt
tail -${PYTEST_TAIL:-60} gpurun_out/pytest_gpu.log
timeout 900 python bench.py --steps ${BENCH_STEPS:-31} --warmup 3 > gpurun_out/bench.log 2>&1; echo "bench rc=$?" | tee -a gpurun_out/summary.txt
tail -3 gpurun_out/bench.log
if [ -n "${ROCPROF}" ]; then
  rm -rf gpurun_out/prof
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 31 --warmup 3 --cpu-frames 0 > $GRAFT_REPO_ROOT/gpurun_out/bench_prof.log 2>&1); echo "rocprof rc=$?" | tee -a gpurun_out/summary.txt
  find gpurun_out/prof -name "*kernel_stats*" | head; 
  f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -30 "$f"
  # keep the merged output small: drop the per-dispatch trace, keep stats
  find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
fi
tail -5 gpurun_out/smoke.log
