#!/bin/bash
# round-3 GPU pass f: consumer-side GroupNorm fold on the small maps — tests, then A/B in the frame (SGAM_GN_FOLD=0 / 1, same build)
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -k "splitk or full_model or trajectory or lockstep or hip_graph or batched or groupnorm or config5 or small_magnitude" > gpurun_out/pytest_r03f.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/pytest_r03f.log
for rep in 1 2 3; do for f in 0 1; do
  echo -n "GN_FOLD=$f rep $rep: "; SGAM_GN_FOLD=$f python bench.py --steps 60 --warmup 5 --no-secondary --cpu-frames 0 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(d['value'], d['ms_per_step'], d['roofline']['kernels_per_frame'], d['roofline']['kernel_time_ms_per_frame'])"
done; done
