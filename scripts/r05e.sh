#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 300 python scripts/step_breakdown.py 2>&1 | tail -8
timeout 300 python scripts/graph_gap.py 2>&1 | tail -6
