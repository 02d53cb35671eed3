#!/bin/bash
# the two oracle-heavy GPU tests with the oracle's thread pool capped at 32 (conftest default on > 64-thread hosts) and uncapped
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
nproc; python -c "import torch; print('torch threads', torch.get_num_threads())"
for t in 32 16 1000; do
T0=$SECONDS; SGAM_TEST_THREADS=$t timeout 900 python -m pytest tests/test_gpu_configs.py -q -k "free_running_32" 2>&1 | tail -1; echo "threads=$t wall=$((SECONDS-T0))s"
done
