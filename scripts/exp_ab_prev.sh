#!/bin/bash
# same-box A/B: the library built from the `_prev/` worktree (a checkout of an earlier commit) against the working tree's.
#   CMDS='python bench.py --dtype fp16 ...;python scripts/shape_time.py ...' scripts/exp_ab_prev.sh
cd $GRAFT_REPO_ROOT
mkdir -p /tmp/ab/prev /tmp/ab/cur
(cd _prev && SGAM_LIB_DIR=/tmp/ab/prev python -m sgam_neurips22_amd.build 2>&1 | grep -E "error")
SGAM_LIB_DIR=/tmp/ab/cur python -m sgam_neurips22_amd.build 2>&1 | grep -E "error"
IFS=';' read -ra CL <<< "${CMDS:-python bench.py --dtype fp16 --steps 60 --warmup 5 --no-secondary --cpu-frames 0 --no-roofline}"
for rep in 1 2; do
  for name in prev cur; do
    echo "== $name (rep $rep)"
    export SGAM_HIP_LIB=/tmp/ab/$name/libsgam_hip.so
    for c in "${CL[@]}"; do eval "$c" 2>/dev/null | grep -E "plan|value|us per" | cut -c1-${CUT:-150}; done
  done
done
