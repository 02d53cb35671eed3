#!/bin/bash
# round 5, call 11: GroupNorm statistics as integer accumulators (SGAM_STATS_ACC=1) with 64 instead of 16 replicas of the record
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
A=$GRAFT_REPO_ROOT/ablib; MAIN=$GRAFT_REPO_ROOT/sgam_neurips22_amd/lib/libsgam_hip.so
fr() { n=$1; m=$2; shift; shift; echo -n "$n $m: "; env "$@" timeout 300 python scripts/h16_frame.py $m 2>&1 | tail -9 | head -1; }
for r in 1 2; do
for m in bf16 f32; do
fr main $m SGAM_HIP_LIB=$MAIN
fr acc16 $m SGAM_HIP_LIB=$MAIN SGAM_STATS_ACC=1
fr acc64 $m SGAM_HIP_LIB=$A/r64/libsgam_hip.so SGAM_STATS_ACC=1 SGAM_STATS_R=64
done; done
