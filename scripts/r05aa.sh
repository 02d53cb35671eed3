#!/bin/bash
# round 5, call 28: the fused AttnBlock front ends fold their input's chunk statistics themselves: tests, frames with the fold off / on
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_h16.py -q -k "three_launches or front_end" 2>&1 | tail -12
timeout 1200 python -m pytest tests/test_gpu_vqgan.py tests/test_gpu_h16.py tests/test_gpu_configs.py -q -k "full_model or 16bit or h16_vs or config2 or parity" 2>&1 | tail -5
fr() { n=$1; m=$2; shift; shift; echo -n "$n $m: "; env "$@" timeout 300 python scripts/h16_frame.py $m 2>&1 | tail -9 | head -${HEADN:-1}; }
for r in 1 2 3; do
for m in f32 bf16; do
fr table $m SGAM_ATTN_FOLD=0
fr fold $m SGAM_ATTN_FOLD=1
done; done
