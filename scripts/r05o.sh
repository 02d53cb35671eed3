#!/bin/bash
# round 5, call 15: (1) fix-up + accumulator statistics with the replica chosen by tile; (2) pose sets / resume on the device;
# (3) 16-bit attention with normalised fp16 partial outputs: attention + agreement tests, bf16 / fp16 frames against the previous build
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
SGAM_XFIXUP=1 timeout 600 python -m pytest tests/test_gpu_fixup.py -q 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_pose_sets.py -q 2>&1 | tail -15
timeout 1200 python -m pytest tests/test_gpu_h16.py tests/test_gpu_configs.py tests/test_gpu_vqgan.py tests/test_gpu_ops.py -q -k "attention or attn or 16bit or h16" 2>&1 | tail -15
A=$GRAFT_REPO_ROOT/ablib; MAIN=$GRAFT_REPO_ROOT/sgam_neurips22_amd/lib/libsgam_hip.so
fr() { n=$1; m=$2; shift; shift; echo -n "$n $m: "; env "$@" timeout 300 python scripts/h16_frame.py $m 2>&1 | tail -9 | head -${HEADN:-1}; }
for r in 1 2 3; do
for m in bf16 fp16; do
fr pre $m SGAM_HIP_LIB=$A/pre_attn/libsgam_hip.so
fr main $m SGAM_HIP_LIB=$MAIN
done; done
HEADN=9 fr pre bf16 SGAM_HIP_LIB=$A/pre_attn/libsgam_hip.so
HEADN=9 fr main bf16 SGAM_HIP_LIB=$MAIN
for L in $A/pre_attn/libsgam_hip.so $MAIN; do SGAM_HIP_LIB=$L python - <<'P'
import torch, sys, os
sys.path.insert(0, os.getcwd())
from sgam_neurips22_amd import ops
for dt in (torch.bfloat16, torch.float16):
    torch.manual_seed(0)
    qkv = (torch.randn(4096, 768, device="cuda") * 1.5).to(dt)
    ref = torch.softmax((qkv[:, :256].double() @ qkv[:, 256:512].double().T) / 16.0, -1) @ qkv[:, 512:].double()
    out = ops.attention_h16(qkv, 256, 1 / 16.0)
    rec, br = ops.kernel_timeline(lambda: ops.attention_h16(qkv, 256, 1 / 16.0))
    print(os.environ["SGAM_HIP_LIB"][-30:], dt, "max err", float((out.double() - ref).abs().max()), [(r[0][:28], round(1e3 * (r[1] - br), 2)) for r in rec])
P
done
