#!/bin/bash
# round 4: scheduling-barrier variants of the 16-bit halo kernel's slab body (SGAM_HSB 0 / 1 / 2: libraries built in the build container),
# layer timing + bf16 frame; then the new TSDF extraction / export tests
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for rep in 1 2; do for v in hsb0 cur hsb2; do
  [ $v = cur ] && export SGAM_HIP_LIB=$GRAFT_REPO_ROOT/sgam_neurips22_amd/lib/libsgam_hip.so || export SGAM_HIP_LIB=$GRAFT_REPO_ROOT/sgam_neurips22_amd/lib_$v/libsgam_hip.so
  echo "== $v rep $rep"
  python scripts/h16_layer_time.py 1 bf16 2>/dev/null | tail -1 | cut -c60-230
  [ $rep = 1 ] && python scripts/h16_layer_time.py 8 bf16 2>/dev/null | tail -1 | cut -c60-230
  [ $rep = 1 ] && python scripts/h16_layer_time.py 8 bf16 128 128 64 2>/dev/null | tail -1 | cut -c1-230
  python scripts/h16_frame.py bf16 40 2>/dev/null | head -4
  [ $rep = 1 ] && python scripts/h16_frame.py bf16 12 8 2>/dev/null | head -4
done; done
unset SGAM_HIP_LIB
timeout 900 python -m pytest tests/test_gpu_tsdf.py -m gpu -q -x --timeout=600 2>&1 | tail -8
