#!/bin/bash
# round-5 evidence set of one build (stamped with its commit): smoke, the whole GPU suite (+ the experimental tests), the driver-style
# bench line, rocprofv3 kernel stats (f32 split + fp16), in-frame PMC for the three modes
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
export SGAM_COMMIT=${SGAM_COMMIT:-1e988ac}
TAG=${TAG:-r05}
timeout 600 python __graft_entry__.py smoke > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/${TAG}_smoke.log
T0=$SECONDS; timeout 1800 python -m pytest tests -m gpu -q --timeout=900 > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$? wall=$((SECONDS-T0))s"; tail -4 gpurun_out/${TAG}_pytest_gpu.log
SGAM_TEST_EXPERIMENTAL=1 timeout 1500 python -m pytest tests -m experimental -q --timeout=900 > gpurun_out/${TAG}_pytest_experimental.log 2>&1; echo "experimental rc=$?"; tail -3 gpurun_out/${TAG}_pytest_experimental.log
timeout 1200 python bench.py > gpurun_out/${TAG}_bench.log 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"; cp bench_extra.json gpurun_out/${TAG}_bench_extra.json 2>/dev/null; cut -c1-400 gpurun_out/${TAG}_bench.log
bash scripts/prof_stats.sh 2>&1 | tail -4
for m in f32 fp16 bf16; do MODE=$m STEPS=6 bash scripts/pmc_frame.sh 2>&1 | head -6; done
