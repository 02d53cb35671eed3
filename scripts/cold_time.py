#!/usr/bin/env python
"""COLD-cache duration of one conv shape under given plans: every timed launch is preceded by a 1 GiB fill (evicts the L2s
and the memory-side cache), i.e. weights AND activations arrive from HBM like inside a frame, where a layer's operands were
last touched ~3 ms earlier.  Uses the library's own per-kernel event brackets (sgam_prof_*).
   python scripts/cold_time.py "f32x|B1|16x16x512|16x16|N512|k3x3s1u0" 32,32,1 64,128,16"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch  # noqa: E402

from sgam_neurips22_amd import _lib, ops, testing, tune  # noqa: E402
from sgam_neurips22_amd._lib import ConvDesc  # noqa: E402

key = sys.argv[1]
dt, B, Hi, Wi, Cin, Ho, Wo, N, KH, KW, stride, ups = tune._parse(key)
x = testing.seeded_tensor("tune.x", (B * Hi * Wi, Cin)).cuda()
K = KH * KW * Cin
w = ops.split_rows((testing.seeded_tensor("tune.w", (N, K)) * 0.03).cuda(), 1024.0)
out = torch.empty((B * Ho * Wo, N), device="cuda")
flush = torch.empty((1 << 28,), device="cuda", dtype=torch.float32)
pad = (KH // 2) if stride == 1 else 0
base = dict(B=B, Hi=Hi, Wi=Wi, Cin=Cin, Ho=Ho, Wo=Wo, N=N, KH=KH, KW=KW, stride=stride, pad_t=pad, pad_l=pad,
            upsample2x=ups, lda=Cin, ldb=K, ldc=N, ldr=0, n_valid=N, bias_per_row=0)
lib = _lib.load()
for plan in sys.argv[2:]:
    bm, bn, ks = map(int, plan.split(","))
    d = ConvDesc(**base, plan_bm=bm, plan_bn=bn, plan_ksplit=ks)
    nb = lib.sgam_conv2d_f32x_workspace_bytes(ctypes.byref(d))
    ws = torch.empty((max(nb, 16),), device="cuda", dtype=torch.uint8)

    def run():
        lib.sgam_conv2d_nhwc_f32x(ctypes.byref(d), ops._p(x), 1.0, ops._p(w.planes), float(w.scale), None, None, ops._p(out),
                                  ops._p(ws), nb, ops._stream())

    def med(fn):
        fn()
        recs, br = ops.kernel_timeline(fn)
        per = {}
        for name, ms, *_ in recs:
            per.setdefault(name, []).append(ms - br)
        return {k: sorted(v)[len(v) // 2] * 1e3 for k, v in per.items()}

    def cold():
        for _ in range(12):
            flush.fill_(1.0)
            run()

    def warm_w():                     # weights touched after the flush (memory-side cache / some L2s), activations cold
        for _ in range(12):
            flush.fill_(1.0)
            w.planes.view(torch.int32).sum()
            run()

    def hot():
        for _ in range(12):
            run()
    res = {n: med(f) for n, f in (("COLD", cold), ("WARMW", warm_w), ("HOT", hot))}
    print(f"{key} plan {plan}: " + "   ".join(
        f"{n} {sum(v.values()):6.1f} us (" + " ".join(f"{k.split('<')[0][-14:]} {t:.1f}" for k, t in v.items()) + ")" for n, v in res.items()))
