#!/bin/bash
# round 4, first call: smoke, the new world-1 RCCL tests + ABI-sensitive tests, the driver-style bench (ONE compact line + side file)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
timeout 1500 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_vqgan.py tests/test_gpu_lockstep.py -m gpu -q -x --timeout=600 2>&1 | tail -5
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04a_bench.log 2> gpurun_out/r04a_bench.err; echo "bench rc=$?"
wc -c gpurun_out/r04a_bench.log; tail -c 3500 gpurun_out/r04a_bench.log; tail -5 gpurun_out/r04a_bench.err
cp bench_extra.json gpurun_out/r04a_bench_extra.json
