"""Every launch of one eager VQGAN forward (256^2 GoogleEarth, B = 1) IN ORDER with its shape and duration: where launch sequences
without a cross-workgroup dependency sit (what the fused AttnBlock front ends of round 5 were read from).
   python scripts/frame_sequence.py [f32|fp16|bf16]"""
import sys; sys.path.insert(0, "/root/repo")
import torch
from sgam_neurips22_amd import testing, ops
from sgam_neurips22_amd.config import default_params
from sgam_neurips22_amd.generative_sensing_module.model import VQModel
dt = sys.argv[1] if len(sys.argv) > 1 else "f32"
p = default_params("google_earth"); m = VQModel(**p)
sd = testing.synthetic_state_dict(m.state_dict(), seed=0)
sd["quantize.embedding.weight"] = testing.codebook_from_stats(0.0, 0.5, p["n_embed"], 256, 1)
m.load_state_dict(sd); m = m.cuda().eval(); m.set_compute_dtype(dt)
x, em = testing.rect_hole_input(1, 256, 256, seed=40)
x, em = x.cuda(), em.cuda()
def one():
    with torch.no_grad(), m.eager():
        m(x, extrapolation_mask=em)
one(); one()
recs, br = ops.kernel_timeline(one)
tot = 0.0
for i, (name, ms_, fl, by, shp) in enumerate(recs):
    t = max(ms_ - br, 0.0) * 1e3
    tot += t
    print(f"{i:3d} {t:7.1f} us  {name[:58]:58s} {shp}")
print(f"{len(recs)} launches, {tot / 1e3:.3f} ms")
