#!/bin/bash
# A/B of TSDF build variants (ablib/<name>/libsgam_hip.so, built with SGAM_LIB_DIR + SGAM_TSDF_* by sgam_neurips22_amd.build) on ONE
# box: per-kernel durations of the conditioning launches (scripts/rgbd_step_breakdown.py's last line), product library first and last
R=$GRAFT_REPO_ROOT
for v in product "$@" product; do
  if [ "$v" = product ]; then unset SGAM_HIP_LIB; else export SGAM_HIP_LIB=$R/ablib/$v/libsgam_hip.so; fi
  echo "== $v: $(python $R/scripts/rgbd_step_breakdown.py 2>&1 | tail -1)"
done
