#!/bin/bash
# fused attention: two counted barriers per trip (staging a whole trip ahead) against the previous commit's one-barrier loop
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_vqgan.py tests/test_gpu_configs.py -m gpu -x -q -k "attention or attn" 2>&1 | tail -4
CMDS='python scripts/attn_time.py 4096 fused;python scripts/attn_time.py 16384 fused;python bench.py --steps 60 --warmup 5 --no-secondary --cpu-frames 0 --no-roofline' CUT=110 bash scripts/exp_ab_prev.sh
