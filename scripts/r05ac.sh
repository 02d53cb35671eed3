#!/bin/bash
# round 5, call 32: where the split-fp32 flash kernel's time outside its loop goes: no loop (16), no partial-O stores (32), neither (48)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for v in default attnab16 attnab32 attnab48; do
  if [ $v = default ]; then python scripts/attn_flash_ablate.py; else SGAM_HIP_LIB=$GRAFT_REPO_ROOT/ablib/$v/libsgam_hip.so python scripts/attn_flash_ablate.py; fi
done 2>&1 | grep -v amdgpu.ids
