#!/usr/bin/env python
"""The 16-bit halo kernel on the dominant layer as the frame runs it — GroupNorm + swish fused into the staging, bias, optional
residual, output statistics — timed two ways: HOT (one launch replayed back to back from a HIP graph: operands L2 / MALL
resident) and CHAIN (six layers with their own weights ping-ponging three activation buffers behind a 1 GiB flush per
round, replayed from a graph: operands arrive the way they do inside a frame).
   python scripts/h16_layer_time.py [B=1] [dtype=bf16] [HW=256] [C=128] [bm=128] [swish=1]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch  # noqa: E402

from sgam_neurips22_amd import _lib, ops, testing  # noqa: E402
from sgam_neurips22_amd._lib import ConvDesc  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dtn = sys.argv[2] if len(sys.argv) > 2 else "bf16"
HW = int(sys.argv[3]) if len(sys.argv) > 3 else 256
C = int(sys.argv[4]) if len(sys.argv) > 4 else 128
bm = int(sys.argv[5]) if len(sys.argv) > 5 else 128
sw = int(sys.argv[6]) if len(sys.argv) > 6 else 1          # 0: GroupNorm without swish (bounds what a free swish would buy)
dt = ops.DTYPES[dtn]
dev = "cuda"
lib = _lib.load()
NL = 6
xs = [testing.seeded_tensor(f"hl.x{i}", (B * HW * HW, C)).to(dev).to(dt) for i in range(3)]
ws = [(torch.randn((C // 32, 9 * C // 32, 128, 8), device=dev) * 0.03).to(dt) for _ in range(NL)]
bias = torch.zeros(C, device=dev)
gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
mr = torch.tensor([0.0, 1.0], device=dev).repeat(B * 32).contiguous()
d = ConvDesc(B=B, Hi=HW, Wi=HW, Cin=C, Ho=HW, Wo=HW, N=C, KH=3, KW=3, stride=1, pad_t=1, pad_l=1, upsample2x=0, lda=C, ldb=9 * C,
             ldc=C, ldr=C, n_valid=C, bias_per_row=0, plan_bm=bm, plan_bn=128, plan_ksplit=1)
chunks = lib.sgam_conv2d_h16_stats_chunks(ctypes.byref(d))
part = torch.empty((B, max(chunks, 1), 32, 2), device=dev, dtype=torch.float64)
gf = 2.0 * B * HW * HW * C * 9 * C / 1e9


def launch(x, w, res, out):
    rc = lib.sgam_conv2d_halo_nhwc_h16(ctypes.byref(d), ops.H16[dt], ops._p(x), ops._p(mr), ops._p(gamma), ops._p(beta), sw, ops._p(w),
                                       ops._p(bias), ops._p(res) if res is not None else None, ops._p(out), 0, ops._p(part), None, 0,
                                       ops._stream())
    assert rc == 0, rc


def graph_time(fn, reps):
    fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best * 1e3


hot_plain = graph_time(lambda: launch(xs[0], ws[0], None, xs[1]), 40)
hot_res = graph_time(lambda: launch(xs[0], ws[0], xs[2], xs[1]), 40)


def chain():
    # a ResnetBlock pair per two layers: conv1 (x -> h), conv2 (h -> x' + x)
    for i in range(NL):
        a, b_, c = xs[i % 3], xs[(i + 1) % 3], xs[(i + 2) % 3]
        launch(a, ws[i], c if i % 2 else None, b_)


chain_us = graph_time(chain, 8) / NL
# cold: the library's per-kernel event brackets around single launches behind a 1 GiB flush
flush = torch.empty((1 << 28,), device=dev, dtype=torch.float32)


def cold():
    for i in range(8):
        flush.fill_(1.0)
        launch(xs[i % 3], ws[i % NL], xs[(i + 2) % 3], xs[(i + 1) % 3])


cold()
recs, br = ops.kernel_timeline(cold)
cs = sorted(ms - br for name, ms, *_ in recs if "halo" in name)
cold_us = cs[len(cs) // 2] * 1e3
name = next(n for n, *_ in recs if "halo" in n)
print(f"{name} B={B} {dtn} {HW}x{HW}x{C} bm={bm} sw={sw} HPF={os.environ.get('SGAM_HPF', '0')}: hot {hot_plain:6.1f} us ({gf / hot_plain / 1e-3 / 1e3:6.1f} TF/s)  "
      f"hot+res {hot_res:6.1f}  chain {chain_us:6.1f} ({gf / chain_us / 1e-3 / 1e3:6.1f} TF/s = {gf / chain_us / 2.5:.3f} of 2500)  cold {cold_us:6.1f}")
