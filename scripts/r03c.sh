#!/bin/bash
# round-3 GPU pass c: batched attention + lockstep tests, stagger A/B, lockstep plan tuning, per-launch timeline
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x --timeout=900 -k "attention or lockstep or config5 or batched or attn" > gpurun_out/pytest_r03c.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/pytest_r03c.log
SGAM_DUMP_TIMELINE=gpurun_out/tl_r03c_f32.tsv timeout 600 python bench.py --steps 20 --warmup 3 --no-secondary --cpu-frames 0 2>/dev/null | cut -c1-120
VARIANTS="base: stg2:SGAM_XSTAGGER=2 stg4:SGAM_XSTAGGER=4 stg6:SGAM_XSTAGGER=6" BENCH=1 timeout 1500 bash scripts/exp_ab.sh 2>&1 | grep -v "^$" | tail -60 > gpurun_out/stagger_ab.log; cat gpurun_out/stagger_ab.log
timeout 1500 python -m sgam_neurips22_amd.tune --dtypes f32,fp16,bf16 --configs 256x4,256x8 --out gpurun_out/tuned_b48.json > gpurun_out/tune_b48.log 2>&1; echo "tune rc=$?"; tail -5 gpurun_out/tune_b48.log
