#!/bin/bash
# round 5: the merge of the attention's key ranges with its number of ranges as a template argument (all loads in flight), and the
# projection fused behind it (ABI v8; SGAM_ATTN_PROJ=0/1 on one build) — tests, f32 frames, per-kernel times
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_h16.py -x -q -k "attention or attn" 2>&1 | tail -2
fr() { n=$1; m=$2; shift; shift; echo -n "$n $m: "; env "$@" timeout 300 python scripts/h16_frame.py $m 2>&1 | tail -9 | head -${L:-1}; }
for r in 1 2 3; do fr proj1 f32 SGAM_ATTN_PROJ=1; fr proj0 f32 SGAM_ATTN_PROJ=0; done
SGAM_ATTN_PROJ=1 timeout 300 python scripts/frame_timeline.py f32 1 2>&1 | grep -i "attn\|launches" | cut -c1-120
