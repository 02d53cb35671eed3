#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --timeout=900 -k "fused_into_halo or full_model or conv2d_matches or config5 or lockstep or smoke or hip_graph or tile_plans" > gpurun_out/pytest_r03k.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/pytest_r03k.log | cut -c1-200
CMDS='python bench.py --steps 60 --warmup 5 --no-secondary --cpu-frames 0 --no-roofline' CUT=130 bash scripts/exp_ab_prev.sh 2>&1 | grep -E "==|value" | sed 's/"unit".*//' 
