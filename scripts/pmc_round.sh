#!/bin/bash
# PMC passes (separate rocprofv3 --pmc runs per counter group) for the two halo instantiations of the split-fp32 path and
# the 16-bit halo kernel, each on its dominant layer with GroupNorm fused -> gpurun_out/pmc_*/ ; fold with pmc_to_json.py
cd $GRAFT_REPO_ROOT
PMC_DIR=pmc_h128 SHAPE=1,128,128,256,256,3 MICRO_ARGS="--norm" bash scripts/pmc_conv.sh > gpurun_out/pmc_h128.log 2>&1
PMC_DIR=pmc_h64 SHAPE=1,128,128,128,128,3 MICRO_ARGS="--norm" bash scripts/pmc_conv.sh > gpurun_out/pmc_h64.log 2>&1
PMC_DIR=pmc_h16 SHAPE=1,128,128,256,256,3 MICRO_ARGS="--norm --dtype fp16" bash scripts/pmc_conv.sh > gpurun_out/pmc_h16.log 2>&1
python scripts/conv_micro.py --shape 1,128,128,256,256,3 --norm --reps 50
python scripts/conv_micro.py --shape 1,128,128,128,128,3 --norm --reps 50
python scripts/conv_micro.py --shape 1,128,128,256,256,3 --norm --dtype fp16 --reps 50
tail -4 gpurun_out/pmc_h128.log
