#!/bin/bash
# 16-bit fused attention with explicit fragment prefetch (ring of four, three steps ahead) against the compiler-placed reads
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_h16.py tests/test_gpu_configs.py tests/test_gpu_ops.py -m gpu -x -q -k "attention or attn" 2>&1 | tail -3
CMDS='python scripts/attn_time_h16.py 4096 bf16 | grep fused;python scripts/attn_time_h16.py 16384 bf16 | grep fused;python bench.py --dtype bf16 --steps 60 --warmup 5 --no-secondary --cpu-frames 0 --no-roofline' CUT=110 bash scripts/exp_ab_prev.sh
