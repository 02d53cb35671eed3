#!/usr/bin/env python
"""16-bit halo kernel, 256-row tile (one workgroup per CU, nine-set weight ring) against the 128-row tile on the dominant layer:
same results bit for bit (same accumulation order), and the timing of both."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from sgam_neurips22_amd import ops, testing
dt = torch.float16
x = testing.seeded_tensor("h256.x", (1, 256, 256, 128), 1.3, 0.4).to(dt).cuda()
w = testing.seeded_tensor("h256.w", (128, 128, 3, 3), scale=(1.0 / (128 * 9)) ** 0.5).cuda()
b = testing.seeded_tensor("h256.b", (128,), scale=0.1).cuda()
g, bt = (1 + 0.1 * testing.seeded_tensor("h256.g", (128,))).cuda(), (0.1 * testing.seeded_tensor("h256.bt", (128,))).cuda()
wp = ops.pack_conv_weight(w, dtype=dt)
wp._sgam_frag_src = w
key = "float16|B1|256x256x128|256x256|N128|k3x3s1u0"
outs = {}
for plan in ((128, 128, 1), (256, 128, 1)):
    ops.PLAN_CACHE[key] = plan
    outs[plan] = ops.conv2d_nhwc(x, wp, b, cout=128, kh=3, kw=3, pad_t=1, pad_l=1, norm=(g, bt, True, 32, 1e-6))
    st = ops.groupnorm_meanrstd(outs[plan])
    outs[plan] = (outs[plan], st)
a, c = outs[(128, 128, 1)], outs[(256, 128, 1)]
print("outputs bit-equal:", torch.equal(a[0], c[0]), " stats max diff:", (a[1] - c[1]).abs().max().item())
