#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
A=$GRAFT_REPO_ROOT/ablib; MAIN=$GRAFT_REPO_ROOT/sgam_neurips22_amd/lib/libsgam_hip.so
timeout 600 python scripts/h16_pc_check.py 2>&1 | tail -3 | cut -c1-200
T0=$(date +%s); timeout 1800 python -m pytest tests -m gpu -q --timeout=900 > gpurun_out/r05l_pytest_gpu.log 2>&1; echo "pytest rc=$? wall $(($(date +%s) - T0)) s"; tail -4 gpurun_out/r05l_pytest_gpu.log
fr() { echo -n "$1 $2: "; SGAM_HIP_LIB=$3 timeout 300 python scripts/h16_frame.py $2 2>&1 | tail -9 | head -${4:-5}; }
for r in 1 2; do fr main bf16 $MAIN 3; fr fd2 bf16 $A/fd2/libsgam_hip.so 3; fr fd3 bf16 $A/fd3/libsgam_hip.so 3; done
