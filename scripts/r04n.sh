#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_vqgan.py -x -q -k "launch_free or full_model_parity" 2>&1 | tail -8
timeout 900 python -m pytest tests/test_gpu_fixup.py tests/test_gpu_ops.py tests/test_gpu_h16.py -x -q 2>&1 | tail -5
