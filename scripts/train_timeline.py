#!/usr/bin/env python
"""Per-kernel timeline of one training step (VQGANTrainer, 256x256 GoogleEarth model, LPIPS + discriminator on): where the
update's time goes.  Uses the library's own event brackets (sgam_prof_*)."""
import collections, os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
import bench
from sgam_neurips22_amd import ops, testing, training
from sgam_neurips22_amd.generative_sensing_module.modules.losses.vqperceptual import VQLPIPSWithDiscriminator
dev = "cuda"
m = bench.build_model(dev)[0]
cfg = VQLPIPSWithDiscriminator(disc_start=0, perceptual_weight=1.0, disc_in_channels=4, disc_weight=0.8, use_discriminative_loss=True).to(dev).train()
cfg.perceptual_loss.load_state_dict({k: v.to(dev) for k, v in testing.synthetic_vgg_state_dict(cfg.perceptual_loss.state_dict(), seed=4).items()})
tr = training.VQGANTrainer(m, cfg, phase="conditional_generation", lr=4.5e-6)
x, mk = testing.rect_hole_input(1, 256, 256, seed=9)
xd = testing.seeded_tensor("bench.train.dst", (1, 4, 256, 256), scale=0.5).clamp(-1, 1).to(dev)
x, mk = x.to(dev), mk.to(dev)
tr.step(x, xd, mk)
recs, br = ops.kernel_timeline(lambda: tr.step(x, xd, mk))
agg = collections.defaultdict(lambda: [0, 0.0])
for name, ms, *_ in recs:
    agg[name][0] += 1
    agg[name][1] += max(ms - br, 0.0)
tot = sum(v[1] for v in agg.values())
print(f"{len(recs)} launches, {tot:.2f} ms of kernel time")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{v[1]:8.3f} ms {v[0]:5d}  {k[:90]}")
