#!/bin/bash
# 16-bit fused attention at <= 256 workgroups: eight wavefronts (two groups over the halves of a key range, merged in LDS) against four
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_h16.py tests/test_gpu_configs.py tests/test_gpu_ops.py tests/test_gpu_lockstep.py -m gpu -q -x -k "attention or attn or fp16 or bf16 or 16bit" 2>&1 | tail -3
for rep in 1 2 3; do for f in 0 1; do
  echo -n "W8=$f: "; SGAM_ATTN_H16_W8=$f python scripts/attn_time_h16.py 4096 bf16 2>/dev/null | grep fused | cut -c1-60
  for dt in bf16 fp16; do echo -n "W8=$f $dt: "; SGAM_ATTN_H16_W8=$f python bench.py --dtype $dt --steps 60 --warmup 5 --no-secondary --cpu-frames 0 --no-roofline 2>/dev/null | cut -c60-110; done
done; done
