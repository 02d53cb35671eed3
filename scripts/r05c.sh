#!/bin/bash
# round 5, call 3: in-frame counters of the three modes on ONE build (stamped with its commit), frame timelines per (kernel, shape),
# f32 / bf16 frames of reference against default build
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
export SGAM_COMMIT=${SGAM_COMMIT:-d24c562}
A=$GRAFT_REPO_ROOT/ablib
fr() { echo -n "$1 $2: "; SGAM_HIP_LIB=$3 timeout 300 python scripts/h16_frame.py $2 2>&1 | tail -9 | head -${4:-5}; }
MAIN=$GRAFT_REPO_ROOT/sgam_neurips22_amd/lib/libsgam_hip.so
for r in 1 2; do fr ref f32 $A/ref/libsgam_hip.so; fr main f32 $MAIN; done
fr ref bf16 $A/ref/libsgam_hip.so 4; fr main bf16 $MAIN 4
for m in f32 bf16; do timeout 300 python scripts/frame_timeline.py $m 1 > gpurun_out/r05_timeline_${m}_b1.txt 2>&1; tail -3 gpurun_out/r05_timeline_${m}_b1.txt; done
for m in f32 bf16 fp16; do MODE=$m bash scripts/pmc_frame.sh 2>&1 | tail -8; done
