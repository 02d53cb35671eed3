#!/bin/bash
# round-3 GPU pass d: batched attention (<= 2048 keys / range) + config5 + lockstep tests, bench with the tuned lockstep plans
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -k "attention or lockstep or config5 or batched or attn" > gpurun_out/pytest_r03d.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/pytest_r03d.log
timeout 1200 python bench.py --steps 31 --warmup 3 > gpurun_out/bench_r03d.log 2> gpurun_out/bench_r03d.err; echo "bench rc=$?"; tail -c 400 gpurun_out/bench_r03d.err
python - <<'PY'
import json
for ln in open("gpurun_out/bench_r03d.log"):
    if ln.startswith("{"):
        d=json.loads(ln); print("value",d["value"],"ms",d["ms_per_step"], "traffic", d["roofline"].get("traffic"), d["roofline"].get("mfma_busy_frac"))
        for k,v in (d.get("lockstep_scenes") or {}).items():
            if isinstance(v,dict): print(k, v["value"], v["ms_per_round"], v["roofline"]["kernel"], v["roofline"]["frac"], v["roofline"]["kernels_per_frame"])
        print("conc", d["concurrent_scenes"]["value"]); print({k:(v["value"], v.get("index_agreement_vs_f32_path")) for k,v in d["throughput_mode"].items() if isinstance(v,dict)})
PY
