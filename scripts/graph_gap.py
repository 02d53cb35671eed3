#!/usr/bin/env python
"""Per-node floor of a HIP graph of dependent kernels on this box: N tiny launches in one captured chain, replayed."""
import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from sgam_neurips22_amd import ops
x = torch.zeros(64, device="cuda")
big = torch.zeros(1 << 22, device="cuda")
for n, t in ((250, x), (250, big)):
    for _ in range(3): t.add_(1.0)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): t.add_(1.0)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    print(f"{n} dependent launches over {t.numel()} floats: {best / n * 1e3:.2f} us per node")
