#!/usr/bin/env python
"""Micro-benchmark of one conv / GEMM shape through the C ABI (for rocprofv3 --pmc passes and A/B timing).
   python scripts/conv_micro.py --shape 1,128,128,256,256,3 --reps 20     (B,Cin,Cout,H,W,k)"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from sgam_neurips22_amd import ops, testing  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="1,128,128,256,256,3")
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--ups", action="store_true")
ap.add_argument("--gn", action="store_true")
ap.add_argument("--norm", action="store_true", help="GroupNorm(+swish) fused into the staging of the split-fp32 / 16-bit halo kernel")
ap.add_argument("--no-swish", action="store_true", help="with --norm: GroupNorm without the swish")
ap.add_argument("--dtype", default="f32")
a = ap.parse_args()
B, Cin, Cout, H, W, k = map(int, a.shape.split(","))
dev = "cuda"
dt = ops.DTYPES[a.dtype]
x = testing.seeded_tensor("micro.x", (B, H, W, Cin)).to(dev).to(dt)
w = ops.pack_conv_weight(testing.seeded_tensor("micro.w", (Cout, Cin, k, k), 0.03).to(dev),
                         dtype="f32x" if (dt == torch.float32 and ops.F32_MODE == "split" and not a.gn) else dt)
b = testing.seeded_tensor("micro.b", (Cout,)).to(dev)
gn = None
if a.gn:
    gn = (ops.groupnorm_stats(x, torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev)), True)
norm = (torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev), not a.no_swish, 32, 1e-6) if a.norm else None
if a.norm:
    x._gn_partials = None
    mr = ops.groupnorm_meanrstd(x)          # statistics once: the timed launches are the conv alone
    if dt in ops.H16:
        w._sgam_frag_src = testing.seeded_tensor("micro.w", (Cout, Cin, k, k), 0.03).to(dev)
_real_meanrstd = ops.groupnorm_meanrstd
if a.norm:
    ops.groupnorm_meanrstd = lambda t, eps=1e-6: mr
for _ in range(3):
    y = ops.conv2d_nhwc(x, w, b, cout=Cout, kh=k, kw=k, pad_t=k // 2, pad_l=k // 2, upsample2x=a.ups, gn=gn, norm=norm)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.reps):
    y = ops.conv2d_nhwc(x, w, b, cout=Cout, kh=k, kw=k, pad_t=k // 2, pad_l=k // 2, upsample2x=a.ups, gn=gn, norm=norm)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.reps
M = y.shape[0] * y.shape[1] * y.shape[2]
fl = 2.0 * M * Cout * k * k * Cin
print(f"shape {a.shape} ups={a.ups} gn={a.gn} norm={a.norm} swish={a.norm and not a.no_swish} {a.dtype}: {ms * 1e3:.1f} us/launch, {fl / ms / 1e9:.1f} TFLOP/s, out {tuple(y.shape)}")
