"""Every launch of one eager VQGAN forward (256^2 GoogleEarth, B = 1) with its shape, aggregated by (kernel, shape): calls,
us per call, share.   python scripts/frame_timeline.py [f32|fp16|bf16] [B]"""
import sys; sys.path.insert(0, "/root/repo")
import collections
import torch
from sgam_neurips22_amd import testing, ops
from sgam_neurips22_amd.config import default_params
from sgam_neurips22_amd.generative_sensing_module.model import VQModel
dt = sys.argv[1] if len(sys.argv) > 1 else "f32"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
p = default_params("google_earth"); m = VQModel(**p)
sd = testing.synthetic_state_dict(m.state_dict(), seed=0)
sd["quantize.embedding.weight"] = testing.codebook_from_stats(0.0, 0.5, p["n_embed"], 256, 1)
m.load_state_dict(sd); m = m.cuda().eval(); m.set_compute_dtype(dt)
xs, ms = zip(*[testing.rect_hole_input(1, 256, 256, seed=40 + i) for i in range(B)])
x, em = torch.cat(xs).cuda(), torch.cat(ms).cuda()
def one():
    with torch.no_grad(), m.eager():
        m(x, extrapolation_mask=em)
one(); one()
recs, br = ops.kernel_timeline(one)
agg = collections.OrderedDict()
for name, ms_, fl, by, shp in recs:
    a = agg.setdefault((name, shp), [0, 0.0, 0.0])
    a[0] += 1; a[1] += max(ms_ - br, 0.0); a[2] += fl
tot = sum(a[1] for a in agg.values())
print(f"{len(recs)} launches, {tot:.3f} ms of kernel time (bracket {br * 1e3:.1f} us subtracted)")
for (name, shp), a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{a[1] / tot * 100:5.1f} %  {a[0]:3d} x {a[1] / a[0] * 1e3:7.1f} us  {a[2] / max(a[1], 1e-9) / 1e9:7.1f} TF/s  {name[:60]:60s} M,N,K,ks={shp}")
