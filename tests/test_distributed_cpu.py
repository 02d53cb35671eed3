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
