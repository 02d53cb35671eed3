#!/bin/bash
# The evidence set of a build, in ONE gpurun call (TAG=r06 bash scripts/round_evidence.sh): smoke(), the GPU suite, rocprofv3 kernel
# stats of the bench (fp32 split + fp16) and of the rgbd_integration branch, in-frame PMC counters of the three modes and of the TSDF
# kernels, and LAST the driver-style bench line reading the counters just collected (so its roofline.counters_commit is this build
# and counters_stale is false).  Everything lands in gpurun_out/ under names starting with $TAG; copy what is to be judged into
# profiles/ (the ${TAG}_pmc_index.json written here is profiles/pmc_index.json).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?run through gpurun}; T=${TAG:-rXX}; O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 600 python __graft_entry__.py smoke > $O/${T}_smoke.log 2>&1; echo "smoke rc=$?"
timeout 1500 python -m pytest tests -m gpu -q > $O/${T}_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/${T}_pytest_gpu.log
bash scripts/prof_stats.sh > $O/${T}_prof_stats.log 2>&1; mv $O/f32_stats.csv $O/${T}_f32split_kernel_stats.csv; mv $O/fp16_stats.csv $O/${T}_fp16_kernel_stats.csv
bash scripts/prof_rgbd.sh > $O/${T}_prof_rgbd.log 2>&1; mv $O/rgbd_stats.csv $O/${T}_rgbd_kernel_stats.csv
for m in f32 fp16 bf16; do MODE=$m bash scripts/pmc_frame.sh > $O/${T}_pmc_frame_$m.log 2>&1; mv $O/pmc_frame_$m.json $O/${T}_pmc_frame_$m.json; done
bash scripts/pmc_rgbd.sh > $O/${T}_pmc_rgbd.log 2>&1; mv $O/pmc_rgbd.json $O/${T}_pmc_frame_rgbd.json
rm -rf $O/pmcf_* $O/pmc_rgbd $O/prof_*
for m in f32 fp16 bf16 rgbd; do cp $O/${T}_pmc_frame_$m.json profiles/; done
python scripts/pmc_index_update.py $T && cp profiles/pmc_index.json $O/${T}_pmc_index.json
timeout 900 python bench.py > $O/${T}_bench.log 2>&1; echo "bench rc=$?"; tail -1 $O/${T}_bench.log | cut -c1-400
cp $O/bench_extra.json $O/${T}_bench_extra.json 2>/dev/null
ls -la $O | grep $T
