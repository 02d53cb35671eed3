#!/bin/bash
# round 5: the new split-fp32 block test with its printed difference, then the whole GPU suite
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_ops.py -q -s -k "three_launches" 2>&1 | grep "attn_block_f32x\|passed\|failed"
T0=$SECONDS; timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > gpurun_out/r05b_pytest_gpu.log 2>&1; echo "pytest rc=$? wall=$((SECONDS-T0))s"; tail -5 gpurun_out/r05b_pytest_gpu.log
