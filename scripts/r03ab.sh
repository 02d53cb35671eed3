#!/bin/bash
# lazy rescale of the attention accumulators (both arithmetic modes) against the rescale-on-any-new-maximum form
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_h16.py tests/test_gpu_configs.py tests/test_gpu_ops.py tests/test_gpu_vqgan.py -m gpu -x -q -k "attention or attn" 2>&1 | tail -3
CMDS='python scripts/attn_time.py 4096 fused;python scripts/attn_time_h16.py 4096 bf16 | grep fused;python scripts/attn_time_h16.py 16384 bf16 | grep fused;python bench.py --steps 60 --warmup 5 --no-secondary --cpu-frames 0 --no-roofline;python bench.py --dtype bf16 --steps 60 --warmup 5 --no-secondary --cpu-frames 0 --no-roofline' CUT=110 bash scripts/exp_ab_prev.sh
