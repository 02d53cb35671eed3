#!/usr/bin/env python
"""Graph-timed GPU duration of one conv shape under given plans (conv + split-K reduce), for kernel experiments.
   python scripts/shape_time.py "f32x|B1|16x16x512|16x16|N512|k3x3s1u0" 64,64,16 64,64,8 ..."""
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch  # noqa: E402

from sgam_neurips22_amd import ops, testing, tune  # noqa: E402
from sgam_neurips22_amd._lib import ConvDesc  # noqa: E402

key = sys.argv[1]
dt, B, Hi, Wi, Cin, Ho, Wo, N, KH, KW, stride, ups = tune._parse(key)
split = dt == "f32x"
dtype = ops.DTYPES["f32" if dt in ("float32", "f32x") else ("bf16" if dt == "bfloat16" else "fp16")]
x = testing.seeded_tensor("tune.x", (B * Hi * Wi, Cin)).cuda().to(dtype)
K = KH * KW * Cin
w = (testing.seeded_tensor("tune.w", (N, K)) * 0.03).cuda().to(dtype)
if split:
    w = ops.split_rows(w, 1024.0)
out = torch.empty((B * Ho * Wo, N), device="cuda", dtype=dtype)
pad = (KH // 2) if stride == 1 else 0
base = dict(B=B, Hi=Hi, Wi=Wi, Cin=Cin, Ho=Ho, Wo=Wo, N=N, KH=KH, KW=KW, stride=stride, pad_t=pad, pad_l=pad,
            upsample2x=ups, lda=Cin, ldb=K, ldc=N, ldr=0, n_valid=N, bias_per_row=0)
gf = 2.0 * B * Ho * Wo * N * K / 1e9
for plan in sys.argv[2:]:
    bm, bn, ks = map(int, plan.split(","))
    t = tune._time(ConvDesc(**base, plan_bm=bm, plan_bn=bn, plan_ksplit=ks), x, w, out, reps=50)
    print(f"{key} plan {plan}: {t * 1e3:7.1f} us  {gf / t / 1e3:6.1f} TFLOP/s" if t else f"{key} plan {plan}: rejected")
