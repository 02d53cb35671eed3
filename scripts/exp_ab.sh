#!/bin/bash
# same-box A/B of kernel build variants: builds each variant into its own directory ON the box, then alternates timings.
#   VARIANTS="name1:ENV=val,ENV2=val name2:..." scripts/exp_ab.sh
cd $GRAFT_REPO_ROOT
VARIANTS=${VARIANTS:-"base: soff0:SGAM_XSOFF=0"}
for v in $VARIANTS; do
  name=${v%%:*}; envs=${v#*:}
  mkdir -p /tmp/ab/$name
  ( IFS=,; for e in $envs; do [ -n "$e" ] && export "$e"; done; SGAM_LIB_DIR=/tmp/ab/$name python -m sgam_neurips22_amd.build 2>&1 | grep -E "error" )
done
for rep in 1 2; do
  for v in $VARIANTS; do
    name=${v%%:*}
    echo "== $name (rep $rep)"
    export SGAM_HIP_LIB=/tmp/ab/$name/libsgam_hip.so
    python scripts/shape_time.py "f32x|B1|256x256x128|256x256|N128|k3x3s1u0" 128,128,1 | grep plan
    python scripts/shape_time.py "f32x|B1|128x128x128|128x128|N128|k3x3s1u0" 64,128,1 | grep plan
    python scripts/shape_time.py "f32x|B1|64x64x256|64x64|N256|k3x3s1u0" 64,128,2 | grep plan
    [ -n "$BENCH" ] && python bench.py --steps 40 --warmup 5 --no-secondary --cpu-frames 0 --no-roofline 2>/dev/null | cut -c60-110
  done
done
