#!/bin/bash
# round-3 GPU pass b: lockstep tests, C=128 attn test, bench with lockstep leg, in-frame PMC for the three modes
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_lockstep.py tests/test_gpu_vqgan.py -m gpu -q -x --timeout=600 -k "lockstep or c128" > gpurun_out/pytest_r03b.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/pytest_r03b.log
timeout 1200 python bench.py --steps 31 --warmup 3 > gpurun_out/bench_r03b.log 2> gpurun_out/bench_r03b.err; echo "bench rc=$?"; tail -c 600 gpurun_out/bench_r03b.err
python - <<'PY'
import json
for ln in open("gpurun_out/bench_r03b.log"):
    if ln.startswith("{"):
        d=json.loads(ln); print("value",d["value"],"ms",d["ms_per_step"])
        for k,v in (d.get("lockstep_scenes") or {}).items():
            if isinstance(v,dict): print(k, v["value"], v["ms_per_round"], v["roofline"]["kernel"], v["roofline"]["frac"], v["roofline"]["frame"])
        print("conc", d["concurrent_scenes"]["value"]); print({k:(v["value"], v.get("index_agreement_vs_f32_path")) for k,v in d["throughput_mode"].items() if isinstance(v,dict)})
PY
for m in f32 fp16 bf16; do MODE=$m STEPS=6 bash scripts/pmc_frame.sh 2>&1 | tail -18; done
