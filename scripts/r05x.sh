#!/bin/bash
# the GPU suite with its slowest tests listed
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
T0=$SECONDS; timeout 1500 python -m pytest tests -m gpu -q --timeout=900 --durations=25 > gpurun_out/r05c_pytest_gpu.log 2>&1; echo "pytest rc=$? wall=$((SECONDS-T0))s"; tail -40 gpurun_out/r05c_pytest_gpu.log
