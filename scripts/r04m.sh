#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_fixup.py -x -q 2>&1 | tail -15
for f in 0 1; do echo "== SGAM_XFIXUP=$f"; SGAM_XFIXUP=$f timeout 300 python scripts/h16_frame.py f32 2>&1 | tail -9; done
