#!/bin/bash
# round 4: the persistent producer / consumer form of the 16-bit 128-row halo kernel — bit-identity against the one-role kernel,
# layer / frame timing A/B (SGAM_HPC=0/1, one build), parity tests; the splat roofline leg on scene-loop geometry
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python scripts/h16_pc_check.py 2>&1 | tail -12
for rep in 1 2; do for v in 0 1; do
  echo "== SGAM_HPC=$v rep $rep"
  SGAM_HPC=$v timeout 300 python scripts/h16_layer_time.py 1 bf16 2>/dev/null | tail -1 | cut -c60-230
  [ $rep = 1 ] && SGAM_HPC=$v timeout 300 python scripts/h16_layer_time.py 8 bf16 2>/dev/null | tail -1 | cut -c60-230
  SGAM_HPC=$v timeout 300 python scripts/h16_frame.py bf16 40 2>/dev/null | head -4
  [ $rep = 1 ] && SGAM_HPC=$v timeout 300 python scripts/h16_frame.py bf16 12 8 2>/dev/null | head -4
done; done
timeout 1200 python -m pytest tests/test_gpu_h16.py tests/test_gpu_configs.py -m gpu -q -x --timeout=600 -k "16bit or h16 or fp16 or bf16 or halo" 2>&1 | tail -5
python - <<'PY' 2>&1 | tail -8
import json, torch, bench
w = bench.warp_roofline(torch.device("cuda", 0))
for k, v in w["cases"].items():
    print(k, {a: v[a] for a in ("form", "us", "tiled_us", "two_pass_global_atomics_us", "achieved", "frac", "kernels_us")})
json.dump(w, open("gpurun_out/r04g_warp_roofline.json", "w"), indent=1)
PY
