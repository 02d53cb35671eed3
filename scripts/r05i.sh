#!/bin/bash
# round 5, call 9: L2 warm-up of the residual tile in the split-fp32 halo kernels (SGAM_XRWARM): parity tests, f32 frame A/B
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
A=$GRAFT_REPO_ROOT/ablib; MAIN=$GRAFT_REPO_ROOT/sgam_neurips22_amd/lib/libsgam_hip.so
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_vqgan.py -x -q -k "parity" 2>&1 | tail -2
fr() { echo -n "$1 $2: "; SGAM_HIP_LIB=$3 timeout 300 python scripts/h16_frame.py $2 2>&1 | tail -9 | head -${4:-5}; }
for r in 1 2 3; do fr main f32 $MAIN 3; fr xrw0 f32 $A/xrw0/libsgam_hip.so 3; done
