#!/bin/bash
# round 4: (1) atomic micro-benchmark for the splat's winner pass; (2) L2 warm-up touches of the 16-bit halo kernel (SGAM_HPF bits) —
# on the layer (hot / chain / cold) and inside the bf16 frame; (3) the world-1 RCCL tests again
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
./scripts/micro/atomic_splat.bin 2>&1 | tail -16
for pf in 0 1 2 4 7 0 7; do
  SGAM_HPF=$pf python scripts/h16_layer_time.py 1 bf16 2>/dev/null | tail -1
done
for pf in 0 7; do SGAM_HPF=$pf python scripts/h16_layer_time.py 8 bf16 2>/dev/null | tail -1; done
for rep in 1 2; do for pf in 0 7 3; do
  SGAM_HPF=$pf python scripts/h16_frame.py bf16 40 2>/dev/null | head -4
done; done
timeout 900 python -m pytest tests/test_gpu_distributed.py -m gpu -q -x --timeout=600 2>&1 | tail -15
