#!/bin/bash
# round 4: (1) the tiled splat — parity tests + the warp roofline leg; (2) SQ counters of the 16-bit halo kernel on its layer (B = 1, B = 8)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_warp.py tests/test_gpu_lockstep.py -m gpu -q -x --timeout=600 2>&1 | tail -8
python - <<'PY' 2>&1 | tail -40
import json, torch, bench
w = bench.warp_roofline(torch.device("cuda", 0))
for k, v in w["cases"].items():
    print(k, {a: v[a] for a in ("us", "two_pass_global_atomics_us", "achieved", "frac", "kernels_us")})
json.dump(w, open("gpurun_out/r04d_warp_roofline.json", "w"), indent=1)
PY
for B in 1 8; do
  PMC_DIR=pmc_h16_b$B CMD="python $GRAFT_REPO_ROOT/scripts/h16_layer_time.py $B bf16" bash scripts/pmc_conv.sh 2>&1 | grep -E "rc=" | tr '\n' ' '
  python scripts/pmc_kernel_counters.py gpurun_out/pmc_h16_b$B conv3x3_h16_halo gpurun_out/r04d_pmc_h16_halo128_b$B.json > /dev/null
  find gpurun_out/pmc_h16_b$B -name "*.csv" -size +2M -delete
done
python - <<'PY'
import json
for B in (1, 8):
    d = json.load(open(f"gpurun_out/r04d_pmc_h16_halo128_b{B}.json"))
    for k, c in d.items():
        print("B", B, k[:70]); print("  ", {a: (round(b, 1) if isinstance(b, float) else b) for a, b in c.items() if a != "derived"}); print("  derived", c.get("derived"))
PY
