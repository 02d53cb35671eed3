#!/bin/bash
# round 5, call 31: the fused front ends leave Q as the flash kernels' B fragments (coalesced loads, no split in the flash prologue):
# tests, f32 / bf16 frames against the previous build, the flash kernels' time
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_h16.py -q -k "three_launches or front_end or attention" 2>&1 | tail -6
timeout 1200 python -m pytest tests/test_gpu_vqgan.py tests/test_gpu_configs.py tests/test_gpu_lockstep.py -q -k "full_model or 16bit or config2 or parity or lockstep" 2>&1 | tail -5
A=$GRAFT_REPO_ROOT/ablib; MAIN=$GRAFT_REPO_ROOT/sgam_neurips22_amd/lib/libsgam_hip.so
fr() { n=$1; m=$2; shift; shift; echo -n "$n $m: "; env "$@" timeout 300 python scripts/h16_frame.py $m 2>&1 | tail -9 | head -${HEADN:-1}; }
for r in 1 2 3; do
for m in f32 bf16; do
fr prev $m SGAM_HIP_LIB=$A/prevq/libsgam_hip.so
fr main $m SGAM_HIP_LIB=$MAIN
done; done
for L in $A/prevq/libsgam_hip.so $MAIN; do SGAM_HIP_LIB=$L timeout 300 python scripts/frame_timeline.py f32 1 2>&1 | grep -i "attn_flash\|attn_qkv" | head -3; done
for L in $A/prevq/libsgam_hip.so $MAIN; do SGAM_HIP_LIB=$L timeout 300 python scripts/frame_timeline.py bf16 1 2>&1 | grep -i "attn_flash\|attn_qkv" | head -3; done
