#!/bin/bash
# phase offset of the CU's second workgroup slot in multi-wave launches: sweep (env, one build)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for st in 0 102 103 104 106; do
  echo -n "STAGGER=$st: "
  SGAM_STAGGER_F32X=$st python scripts/shape_time.py "f32x|B8|256x256x128|256x256|N128|k3x3s1u0" 128,128,1 2>/dev/null | grep plan | awk '{printf "f32x B8 256^2 %s us  ", $4}'
  SGAM_STAGGER_F32X=$st python scripts/shape_time.py "f32x|B8|128x128x128|128x128|N128|k3x3s1u0" 128,128,1 2>/dev/null | grep plan | awk '{printf "f32x B8 128^2 %s us  ", $4}'
  SGAM_STAGGER_H16=$st python scripts/shape_time.py "bfloat16|B8|256x256x128|256x256|N128|k3x3s1u0" 128,128,1 2>/dev/null | grep plan | awk '{printf "bf16 B8 256^2 %s us  ", $4}'
  SGAM_STAGGER_H16=$st python scripts/shape_time.py "bfloat16|B8|128x128x128|128x128|N128|k3x3s1u0" 128,128,1 2>/dev/null | grep plan | awk '{printf "bf16 B8 128^2 %s us\n", $4}'
done
