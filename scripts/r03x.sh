#!/bin/bash
# 16-bit halo kernel: four wavefronts side by side along N (1 x 4: half the weight-fragment bytes through L1, twice the A reads
# from LDS) against the 2 x 2 grid.  Same box, alternating.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for v in 2 1; do mkdir -p /tmp/ab/w$v; SGAM_HWGM=$v SGAM_LIB_DIR=/tmp/ab/w$v python -m sgam_neurips22_amd.build 2>&1 | grep -E "error"; done
SGAM_HIP_LIB=/tmp/ab/w1/libsgam_hip.so timeout 900 python -m pytest tests/test_gpu_h16.py -m gpu -x -q 2>&1 | tail -3
for rep in 1 2; do for v in 2 1; do
  export SGAM_HIP_LIB=/tmp/ab/w$v/libsgam_hip.so
  echo "== WGM=$v (rep $rep)"
  python scripts/shape_time.py "bfloat16|B1|256x256x128|256x256|N128|k3x3s1u0" 128,128,1 | grep plan
  python scripts/shape_time.py "bfloat16|B8|256x256x128|256x256|N128|k3x3s1u0" 128,128,1 | grep plan
  python scripts/shape_time.py "bfloat16|B8|128x128x128|128x128|N128|k3x3s1u0" 128,128,1 | grep plan
  python scripts/shape_time.py "bfloat16|B1|128x128x128|128x128|N128|k3x3s1u0" 64,128,1 | grep plan
  python scripts/shape_time.py "bfloat16|B1|64x64x256|64x64|N256|k3x3s1u0" 64,128,2 | grep plan
  python scripts/shape_time.py "bfloat16|B1|16x16x512|16x16|N512|k3x3s1u0" 64,128,8 | grep plan
  python bench.py --dtype bf16 --steps 60 --warmup 5 --no-secondary --cpu-frames 0 --no-roofline 2>/dev/null | cut -c60-110
done; done
