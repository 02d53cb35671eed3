"""the split-fp32 flash kernel alone (n = 4096, C = 256) under the SGAM_ATTN_ABLATE builds: per-kernel microseconds of one attention
   SGAM_HIP_LIB=ablib/attnab16/libsgam_hip.so python scripts/attn_flash_ablate.py"""
import os, sys; sys.path.insert(0, "/root/repo")
import torch
from sgam_neurips22_amd import ops, testing
C, n = 256, 4096
qkv = testing.seeded_tensor("attn.t", (n, 3 * C)).cuda()
for _ in range(3): ops.attention(qkv, C, C ** -0.5)
best = {}
for _ in range(7):
    recs, br = ops.kernel_timeline(lambda: ops.attention(qkv, C, C ** -0.5))
    for r in recs:
        t = 1e3 * (r[1] - br)
        best[r[0]] = min(best.get(r[0], 1e9), t)
print(os.environ.get("SGAM_HIP_LIB", "default")[-40:], {k[:28]: round(v, 1) for k, v in best.items()})
