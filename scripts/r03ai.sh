#!/bin/bash
# 16-bit mode: group-major split-K combine + consumer-side GroupNorm fold on the 16^2 / 32^2 maps (SGAM_GN_FOLD=0/1 on one build)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_h16.py tests/test_gpu_lockstep.py tests/test_gpu_configs.py -m gpu -q -x -k "h16 or fp16 or bf16 or 16bit or folded" 2>&1 | tail -3
for rep in 1 2 3; do for f in 0 1; do
  for dt in bf16 fp16; do echo -n "FOLD=$f $dt: "; SGAM_GN_FOLD=$f python bench.py --dtype $dt --steps 60 --warmup 5 --no-secondary --cpu-frames 0 --no-roofline 2>/dev/null | cut -c60-110; done
done; done
SGAM_GN_FOLD=1 python scripts/frame_timeline.py bf16 1 2>/dev/null | sed -n 2,3p
