#!/bin/bash
# round 5, call 2: peeled last slabs in BOTH halo families (h16 default on; f32x new) — parity tests on the default build, then
# frames per second + in-frame kernel averages, reference build (round-4 behaviour) against the variants, alternated
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
A=$GRAFT_REPO_ROOT/ablib
echo "== tests on the default build"
timeout 1200 python -m pytest tests/test_gpu_h16.py tests/test_gpu_ops.py tests/test_gpu_fixup.py tests/test_gpu_warp.py -x -q 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_vqgan.py -x -q -k "parity or h16" 2>&1 | tail -4
echo "== h16 bit identity: ref vs default, hrpf0, hrpf2"
timeout 900 python scripts/h16_variant_check.py $A/ref/libsgam_hip.so $GRAFT_REPO_ROOT/sgam_neurips22_amd/lib/libsgam_hip.so $A/hrpf0/libsgam_hip.so $A/hrpf2/libsgam_hip.so 2>&1 | tail -12
fr() { echo -n "$1 $2: "; SGAM_HIP_LIB=$3 timeout 300 python scripts/h16_frame.py $2 2>&1 | tail -9 | head -${4:-5}; }
MAIN=$GRAFT_REPO_ROOT/sgam_neurips22_amd/lib/libsgam_hip.so
echo "== f32 frame"
for r in 1 2; do fr ref f32 $A/ref/libsgam_hip.so; fr main f32 $MAIN; fr xb2 f32 $A/xb2/libsgam_hip.so; done
echo "== bf16 frame"
for r in 1 2; do fr ref bf16 $A/ref/libsgam_hip.so 4; fr main bf16 $MAIN 4; fr hrpf0 bf16 $A/hrpf0/libsgam_hip.so 4; fr hrpf2 bf16 $A/hrpf2/libsgam_hip.so 4; done
echo "== fp16 frame"
fr ref fp16 $A/ref/libsgam_hip.so 4; fr main fp16 $MAIN 4
