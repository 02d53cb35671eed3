#!/usr/bin/env python
"""frames/s of the 256x256 loop in a 16-bit mode + the in-frame average of every halo-kernel instantiation (library timeline):
   python scripts/h16_frame.py [bf16|fp16|f32] [steps=40] [lockstep S=0]"""
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch  # noqa: E402

import bench  # noqa: E402
from sgam_neurips22_amd import distributed as sdist, ops  # noqa: E402
from sgam_neurips22_amd.inference_pipeline import InfiniteSceneGeneration, synthetic_seed_frame  # noqa: E402

dtn = sys.argv[1] if len(sys.argv) > 1 else "bf16"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
S = int(sys.argv[3]) if len(sys.argv) > 3 else 0
dev = torch.device("cuda", 0)
model, sd, p = bench.build_model(dev)
model.set_compute_dtype(dtn)
model.enable_hip_graph(True)
seed = synthetic_seed_frame(bench.DATASET, seed_index=0)
if S:
    ls = sdist.LockstepScenes(model, bench.DATASET, [synthetic_seed_frame(bench.DATASET, seed_index=i) for i in range(S)],
                              output_dim=(steps + 12, 1))
    dt = bench.timed_loop(ls.step, 4, steps)

    def one():
        with model.eager():
            ls.step()
    fps = S * steps / dt
else:
    sc = InfiniteSceneGeneration(model, bench.DATASET, seed_index=0, output_dim=(steps + 12, 1), seed_frame=seed)

    def step():
        sc.one_step_prediction(sc.next_pose(sc.curr)); sc.curr += 1
    dt = bench.timed_loop(step, 4, steps)

    def one():
        with model.eager():
            step()
    fps = steps / dt
one()
recs, br = ops.kernel_timeline(one)
agg = {}
for name, ms, fl, nb, shp in recs:
    a = agg.setdefault(name, [0, 0.0, 0.0])
    a[0] += 1; a[1] += max(ms - br, 0.0); a[2] += fl / 1e9
tot = sum(a[1] for a in agg.values())
tag = " ".join(f"{k}={os.environ[k]}" for k in ("SGAM_HPF", "SGAM_HPERSIST") if k in os.environ)
print(f"{dtn} S={S} {tag}: {fps:8.1f} frames/s  launches {sum(a[0] for a in agg.values())}  kernel time {tot:.3f} ms")
for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:8]:
    peak = 2500.0 if "h16" in name else (833.3 if "f32x" in name else 0)
    fr = f"{a[2] / a[1] / peak:.3f}" if (peak and a[1] > 0) else "  -  "
    print(f"   {name:62s} x{a[0]:3d}  {1e3 * a[1] / a[0]:7.2f} us  frac {fr}")
