#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_training.py -m gpu -q --timeout=900 > gpurun_out/pytest_r03r.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_r03r.log | cut -c1-200
timeout 600 python scripts/train_timeline.py 2>/dev/null | tail -25
