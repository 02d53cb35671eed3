#!/bin/bash
# round 4: GroupNorm scale / shift from an LDS table (no global loads behind the in-loop halo loads) in the 16-bit and split-fp32 halo
# kernels, A / B against the previous commit's library (lib_prev, built in the build container) on one box; then the parity tests
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
P=$GRAFT_REPO_ROOT/sgam_neurips22_amd/lib_prev/libsgam_hip.so; C=$GRAFT_REPO_ROOT/sgam_neurips22_amd/lib/libsgam_hip.so
for rep in 1 2; do for v in prev cur; do
  [ $v = prev ] && export SGAM_HIP_LIB=$P || export SGAM_HIP_LIB=$C
  echo "== $v rep $rep"
  python scripts/h16_layer_time.py 1 bf16 2>/dev/null | tail -1 | cut -c1-230
  [ $rep = 1 ] && python scripts/h16_layer_time.py 8 bf16 2>/dev/null | tail -1 | cut -c1-230
  python scripts/h16_frame.py bf16 40 2>/dev/null | head -5
  python scripts/h16_frame.py f32 40 2>/dev/null | head -5
  [ $rep = 1 ] && python scripts/h16_frame.py bf16 12 8 2>/dev/null | head -4
  [ $rep = 1 ] && python scripts/h16_frame.py f32 12 8 2>/dev/null | head -4
done; done
unset SGAM_HIP_LIB
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_h16.py tests/test_gpu_vqgan.py -m gpu -q -x --timeout=600 2>&1 | tail -6
