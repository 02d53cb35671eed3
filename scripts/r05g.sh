#!/bin/bash
# round 5, call 7: ring of six in the folding form of the 64-row 16-bit tile (SGAM_HNBRF) — bit identity incl. the GNF tests, bf16 frame A/B
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
A=$GRAFT_REPO_ROOT/ablib; MAIN=$GRAFT_REPO_ROOT/sgam_neurips22_amd/lib/libsgam_hip.so
timeout 600 python scripts/h16_variant_check.py $A/ref/libsgam_hip.so $MAIN 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_h16.py tests/test_gpu_vqgan.py -x -q -k "h16 or 16bit or fold or gnf or halo" 2>&1 | tail -3
fr() { echo -n "$1 $2: "; SGAM_HIP_LIB=$3 timeout 300 python scripts/h16_frame.py $2 2>&1 | tail -9 | head -${4:-5}; }
for r in 1 2 3; do fr main bf16 $MAIN 4; fr nbf3 bf16 $A/nbf3/libsgam_hip.so 4; done
