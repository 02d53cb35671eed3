#!/bin/bash
# round-3 GPU pass e: the whole GPU suite, 16-bit plan tuning with the 256-row tile, bench, rocprofv3 kernel stats
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > gpurun_out/pytest_r03e.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/pytest_r03e.log
timeout 900 python -m sgam_neurips22_amd.tune --dtypes fp16,bf16 --configs 256x4,256x8 --out gpurun_out/tuned_h16_b48.json > gpurun_out/tune_h16_b48.log 2>&1; echo "tune rc=$?"; grep -c "(256, 128" gpurun_out/tune_h16_b48.log; grep "(256, 128" gpurun_out/tune_h16_b48.log | head -12
timeout 1200 python bench.py --steps 31 --warmup 3 > gpurun_out/bench_r03e.log 2> gpurun_out/bench_r03e.err; echo "bench rc=$?"
python - <<'PY'
import json
for ln in open("gpurun_out/bench_r03e.log"):
    if ln.startswith("{"):
        d=json.loads(ln); print("value",d["value"],"ms",d["ms_per_step"]); print("train", d["training_step"])
PY
bash scripts/prof_stats.sh 2>&1 | tail -6
