#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
for B in 1 8; do
  PMC_DIR=pmc_pc_b$B CMD="python $GRAFT_REPO_ROOT/scripts/h16_layer_time.py $B bf16" bash scripts/pmc_conv.sh 2>&1 | grep -E "rc=" | tr '\n' ' '
  python scripts/pmc_kernel_counters.py gpurun_out/pmc_pc_b$B conv3x3_h16_pc gpurun_out/r04i_pmc_h16_pc_b$B.json > /dev/null
  find gpurun_out/pmc_pc_b$B -name "*.csv" -size +1M -delete
done
python - <<'PY'
import json
for B in (1, 8):
    d = json.load(open(f"gpurun_out/r04i_pmc_h16_pc_b{B}.json"))
    for k, c in d.items():
        print("B", B, k[:70]); print("  ", {a: (round(b, 1) if isinstance(b, float) else b) for a, b in c.items() if a != "derived"}); print("  derived", c.get("derived"))
PY
