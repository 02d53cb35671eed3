"""CPU: the N>1 path (scene sharding + the end-of-run metric all-gather) with world_size 2 over gloo."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from sgam_neurips22_amd import distributed as sdist
    r, lr, w = sdist.init(backend="gloo")
    scenes = sdist.shard_scenes(5, r, w)
    sdist.barrier()
    g = sdist.gather_metrics(frames=10 * len(scenes), seconds=1.0 + r, checksum=100 + r, device="cpu")
    torch.save({"scenes": scenes, "g": g}, os.path.join(out_dir, f"r{rank}.pt"))
    torch.distributed.destroy_process_group()


def test_scene_sharding_is_a_partition():
    from sgam_neurips22_amd.distributed import shard_scenes
    for world in (1, 2, 3, 8):
        got = sorted(s for r in range(world) for s in shard_scenes(13, r, world))
        assert got == list(range(13))


@pytest.mark.timeout(120)
def test_metric_gather_world2_gloo(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    assert r0["scenes"] == [0, 2, 4] and r1["scenes"] == [1, 3]
    for r in (r0, r1):  # every rank sees the same aggregate: sum of frames / max of seconds
        assert r["g"]["total_frames"] == 50 and r["g"]["max_seconds"] == 2.0 and r["g"]["frames_per_s"] == 25.0
        assert [p[2] for p in r["g"]["per_rank"]] == [100.0, 101.0]


def test_single_process_gather_is_identity():
    from sgam_neurips22_amd.distributed import gather_metrics
    g = gather_metrics(31, 0.5, 7, "cpu")
    assert g["total_frames"] == 31 and g["frames_per_s"] == 62.0 and g["per_rank"] == [(31.0, 0.5, 7.0)]


@pytest.mark.timeout(300)
def test_bench_gpus2_self_launches_two_ranks():
    """`python bench.py --gpus 2` (no torchrun environment) must start 2 ranks by itself; --dry-run takes them through
    scene sharding and the metric all-gather over gloo without touching a GPU."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1",
                        "--dry-run"], capture_output=True, text=True, env=env, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(line) == 1, r.stdout
    out = json.loads(line[0])
    assert out["n_gpus"] == 2 and out["dry_run"] is True and out["scenes_per_rank"] == [0.0, 1.0]
    assert abs(out["value"] - 10 / 1.01) < 1e-9          # 2 ranks x 5 frames / max(1.00, 1.01) s


@pytest.mark.timeout(600)
def test_bench_gpus8_rehearsal_at_the_real_world_size():
    """BASELINE configs[3] (8 scenes on the 8 GPUs of one node) has never met hardware: rehearse everything about it that does
    not need one — `bench.py --gpus 8` self-launches EIGHT ranks through torch.distributed.run, they rendezvous on 127.0.0.1, the
    scene shards of the eight ranks partition the eight scenes, the metric all-gather returns eight records and exactly ONE JSON
    line comes out, carrying n_gpus = 8 and the whole-job aggregate."""
    import json
    import subprocess
    from sgam_neurips22_amd import distributed as sdist
    shards = [sdist.shard_scenes(8, r, 8) for r in range(8)]
    assert sorted(s for sh in shards for s in sh) == list(range(8)) and all(len(sh) == 1 for sh in shards)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "5", "--warmup", "1",
                        "--dry-run"], capture_output=True, text=True, env=env, timeout=560)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(line) == 1, r.stdout
    out = json.loads(line[0])
    assert out["n_gpus"] == 8 and out["dry_run"] is True
    assert out["scenes_per_rank"] == [float(i) for i in range(8)]          # rank r reported scene r: a partition of the 8 scenes
    assert abs(out["value"] - 8 * 5 / 1.07) < 1e-9                         # 8 ranks x 5 frames / max over ranks (1.00 .. 1.07 s)


def test_bench_refuses_a_world_that_is_not_gpus():
    import subprocess
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True,
                       text=True, env=env, timeout=120)
    assert r.returncode != 0 and "--gpus 2" in (r.stderr + r.stdout)


def _grad_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from sgam_neurips22_amd import distributed as sdist
    from sgam_neurips22_amd import testing, training
    from sgam_neurips22_amd.config import default_params
    from sgam_neurips22_amd.generative_sensing_module.model import VQModel
    sdist.init(backend="gloo")
    m = VQModel(**testing.small_train_params(default_params("google_earth")))
    from sgam_neurips22_amd.generative_sensing_module.modules.losses.vqperceptual import VQLPIPSWithDiscriminator
    cfg = VQLPIPSWithDiscriminator(disc_start=0, perceptual_weight=0.0, disc_in_channels=4, use_discriminative_loss=True).train()
    want = {k: v.clone() for k, v in list(m.state_dict().items()) + [("D." + k, v) for k, v in cfg.discriminator.state_dict().items()]}
    for t in list(m.parameters()) + list(cfg.discriminator.parameters()) + list(cfg.discriminator.buffers()):
        t.data.add_(rank)                            # ranks start apart (each process initialises its PatchGAN from its own RNG)
    tr = training.VQGANTrainer(m, cfg, phase="conditional_generation", lr=1e-4)
    # what DDP does at construction: every rank now holds rank 0's parameters and buffers (rank 0 added 0: the originals)
    have = list(m.state_dict().items()) + [("D." + k, v) for k, v in cfg.discriminator.state_dict().items()]
    flat = torch.cat([v.reshape(-1).double() for _, v in have])
    both = [torch.empty_like(flat) for _ in range(world)]
    torch.distributed.all_gather(both, flat)
    synced = tr.synced_tensors > 0 and all(torch.equal(b, both[0]) for b in both) and \
        (rank != 0 or all(torch.equal(v, want[k]) for k, v in have))
    ps = tr.parameters()
    for i, p in enumerate(ps):                       # rank-dependent stand-in gradients: (rank + 1) * (i + 1)
        tr.grads[p] = torch.full(p.shape, float((rank + 1) * (i + 1)))
    nbytes = tr.allreduce_grads()
    ok = all(torch.equal(tr.grads[p], torch.full(p.shape, 1.5 * (i + 1))) for i, p in enumerate(ps))
    extra = [torch.nn.Parameter(torch.zeros(3, 5)), torch.nn.Parameter(torch.zeros(7))]      # a second bucket (the discriminator's)
    dg = {q: torch.full(q.shape, float(rank + 1)) for q in extra}
    ok = ok and tr._allreduce(dg, extra) == 4 * 22 and all(torch.equal(dg[q], torch.full(q.shape, 1.5)) for q in extra)
    torch.save({"ok": ok, "synced": synced, "bytes": nbytes, "n": sum(p.numel() for p in ps)}, os.path.join(out_dir, f"g{rank}.pt"))
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(180)
def test_training_gradient_bucket_allreduce_world2_gloo(tmp_path):
    """the trainer averages the phase's gradients over the ranks through ONE flat bucket (what DDP does
    for the reference's LightningModule); world 2 over gloo, stand-in gradients"""
    port = _free_port()
    mp.spawn(_grad_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        res = torch.load(tmp_path / f"g{r}.pt")
        assert res["ok"] and res["bytes"] == 4 * res["n"] > 0
        assert res["synced"], "every rank must start from rank 0's parameters and buffers (model and discriminator)"
