#!/bin/bash
# round-3 evidence set with the final code: whole GPU suite, smoke, bench, rocprofv3 kernel stats, in-frame PMC (3 modes)
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python __graft_entry__.py smoke > gpurun_out/smoke_${TAG:-r03j}.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke_${TAG:-r03j}.log
timeout 1800 python -m pytest tests -m gpu -q --timeout=900 > gpurun_out/pytest_${TAG:-r03j}.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_${TAG:-r03j}.log
timeout 1200 python bench.py --steps 31 --warmup 3 > gpurun_out/bench_${TAG:-r03j}.log 2> gpurun_out/bench_${TAG:-r03j}.err; echo "bench rc=$?"
bash scripts/prof_stats.sh 2>&1 | tail -4
for m in f32 fp16 bf16; do MODE=$m STEPS=6 bash scripts/pmc_frame.sh 2>&1 | head -8; done
