#!/bin/bash
# round 5, call 8: ring of six in the split-fp32 64-row halo tile (SGAM_XNBR64): parity tests, f32 frame A/B
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
A=$GRAFT_REPO_ROOT/ablib; MAIN=$GRAFT_REPO_ROOT/sgam_neurips22_amd/lib/libsgam_hip.so
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fixup.py -x -q 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_vqgan.py -x -q -k "parity or determin or graph" 2>&1 | tail -3
fr() { echo -n "$1 $2: "; SGAM_HIP_LIB=$3 timeout 300 python scripts/h16_frame.py $2 2>&1 | tail -9 | head -${4:-5}; }
for r in 1 2 3; do fr main f32 $MAIN 4; fr xnb3 f32 $A/xnb3/libsgam_hip.so 4; done
