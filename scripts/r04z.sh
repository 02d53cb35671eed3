#!/bin/bash
# round 4, the round's record: smoke, the full GPU suite (timed), the driver-style bench line + side file, rocprofv3 kernel stats (fp32 split,
# fp16), in-frame PMC passes (fp32, bf16)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2 | tee gpurun_out/r04z_smoke.log
( time timeout 1500 python -m pytest tests -m gpu -q --timeout=600 ) > gpurun_out/r04z_pytest_gpu.log 2>&1; tail -6 gpurun_out/r04z_pytest_gpu.log
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04z_bench.log 2> gpurun_out/r04z_bench.err; echo "bench rc=$? bytes $(wc -c < gpurun_out/r04z_bench.log)"
cp bench_extra.json gpurun_out/r04z_bench_extra.json
bash scripts/prof_stats.sh 2>&1 | tail -4
cp gpurun_out/f32_stats.csv gpurun_out/r04z_f32split_kernel_stats.csv; cp gpurun_out/fp16_stats.csv gpurun_out/r04z_fp16_kernel_stats.csv
for m in f32 bf16; do MODE=$m bash scripts/pmc_frame.sh 2>&1 | tail -3; cp gpurun_out/pmc_frame_$m.json gpurun_out/r04z_pmc_frame_$m.json; done
rm -rf gpurun_out/pmcf_* gpurun_out/prof_f32 gpurun_out/prof_fp16
