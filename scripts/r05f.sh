#!/bin/bash
# round 5, call 6: six-deep weight ring for the whole-K launches of the 64-row 16-bit tile (SGAM_HNBR64=6) in the bf16 / fp16 frame
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
A=$GRAFT_REPO_ROOT/ablib; MAIN=$GRAFT_REPO_ROOT/sgam_neurips22_amd/lib/libsgam_hip.so
timeout 600 python scripts/h16_variant_check.py $MAIN $A/nb64/libsgam_hip.so 2>&1 | tail -4
fr() { echo -n "$1 $2: "; SGAM_HIP_LIB=$3 timeout 300 python scripts/h16_frame.py $2 2>&1 | tail -9 | head -${4:-5}; }
for r in 1 2 3; do fr main bf16 $MAIN 4; fr nb64 bf16 $A/nb64/libsgam_hip.so 4; done
fr main fp16 $MAIN 4; fr nb64 fp16 $A/nb64/libsgam_hip.so 4
