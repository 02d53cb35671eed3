#!/bin/bash
# round 5, call 14: in-kernel split-K fix-up with 32 coherent loads in flight per thread of the last arriver (SGAM_XFIX_LF=32),
# for every split-K launch and for grids of <= 256 / 512 workgroups only (SGAM_XFIXUP_MAXWG); bit identity (test_gpu_fixup.py), f32 frames
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
SGAM_XFIXUP=1 timeout 600 python -m pytest tests/test_gpu_fixup.py -q 2>&1 | tail -40
fr() { n=$1; m=$2; shift; shift; echo -n "$n $m: "; env "$@" timeout 300 python scripts/h16_frame.py $m 2>&1 | tail -9 | head -${HEADN:-1}; }
for r in 1 2 3; do
fr off f32 SGAM_XFIXUP=0
fr fix_all f32 SGAM_XFIXUP=1
fr fix_256 f32 SGAM_XFIXUP=1 SGAM_XFIXUP_MAXWG=256
fr fix_512 f32 SGAM_XFIXUP=1 SGAM_XFIXUP_MAXWG=512
fr fix_lf0 f32 SGAM_XFIXUP=1 SGAM_HIP_LIB=$GRAFT_REPO_ROOT/ablib/lf0/libsgam_hip.so
done
HEADN=9 fr off f32 SGAM_XFIXUP=0
HEADN=9 fr fix_all f32 SGAM_XFIXUP=1
