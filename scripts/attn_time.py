"""Graph-timed fused attention vs the GEMM / softmax / GEMM chain at n tokens, C = 256."""
import sys; sys.path.insert(0, "/root/repo")
import torch
from sgam_neurips22_amd import ops, testing
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
which = sys.argv[2] if len(sys.argv) > 2 else "both"
C = 256
qkv = testing.seeded_tensor("attn.t", (n, 3 * C)).cuda()
scale = C ** -0.5
def chain():
    vt = ops.nhwc_to_nchw(qkv[:, 2 * C:].unsqueeze(0).unsqueeze(0), c=C).view(C, n)
    s = ops.gemm_nt(qkv[:, :C], qkv[:, C:2 * C]); ops.softmax_rows_(s, scale)
    return ops.gemm_nt(s, vt, a_scale=1024.0)
def fused():
    return ops.attention(qkv, C, scale)
for name, f in (("chain", chain), ("fused", fused)):
    if which not in ("both", name): continue
    for _ in range(3): f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10): r = f()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us per attention (n={n})")
if which == "both": print("max |fused - chain|", (fused() - chain()).abs().max().item())
