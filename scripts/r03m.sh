#!/bin/bash
# A/B of the working tree against the previous commit (_prev/ worktree) on one box: frames/s, launches and kernel time per frame
mkdir -p gpurun_out; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p /tmp/ab/prev /tmp/ab/cur
(cd _prev && SGAM_LIB_DIR=/tmp/ab/prev python -m sgam_neurips22_amd.build 2>&1 | grep -E "error")
SGAM_LIB_DIR=/tmp/ab/cur python -m sgam_neurips22_amd.build 2>&1 | grep -E "error"
for rep in 1 2 3; do for name in prev cur; do
  echo -n "$name rep $rep: "
  SGAM_HIP_LIB=/tmp/ab/$name/libsgam_hip.so python bench.py --steps 60 --warmup 5 --no-secondary --cpu-frames 0 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(d['value'], d['ms_per_step'], d['roofline']['kernels_per_frame'], d['roofline']['kernel_time_ms_per_frame'])"
done; done
