#!/bin/bash
# round 4: tiled splat v2 (bins register with tiles; wave-parallel candidates) — parity + roofline leg; TSDF extraction tests
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_warp.py tests/test_gpu_tsdf.py tests/test_gpu_lockstep.py -m gpu -q -x --timeout=600 2>&1 | tail -8
python - <<'PY' 2>&1 | tail -12
import json, torch, bench
w = bench.warp_roofline(torch.device("cuda", 0))
for k, v in w["cases"].items():
    print(k, {a: v[a] for a in ("form", "us", "tiled_us", "two_pass_global_atomics_us", "achieved", "frac", "kernels_us")})
json.dump(w, open("gpurun_out/r04f_warp_roofline.json", "w"), indent=1)
PY
