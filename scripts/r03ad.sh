#!/bin/bash
# generic 16-bit kernel: transposed product + direct epilogue (16-byte stores) + GroupNorm statistics of the output
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_h16.py tests/test_gpu_lockstep.py -m gpu -x -q 2>&1 | tail -4
CMDS='python bench.py --dtype bf16 --steps 60 --warmup 5 --no-secondary --cpu-frames 0 --no-roofline;python bench.py --dtype fp16 --steps 60 --warmup 5 --no-secondary --cpu-frames 0 --no-roofline' CUT=110 bash scripts/exp_ab_prev.sh
python scripts/frame_timeline.py bf16 1 2>/dev/null | head -3
