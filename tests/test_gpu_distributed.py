"""The multi-GPU path's unknowns that ONE GPU can exercise: RCCL initialisation from a torchrun-style
environment, the metric all-gather and the trainer's broadcast / gradient all-reduce on DEVICE tensors over the `nccl`
backend, teardown — and `bench.py` under the exact launcher command line `distributed.self_launch` / the driver build
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 ...`, `HSA_ENABLE_IPC_MODE_LEGACY=0`).  Each case runs in
its own process so that the pytest process never owns a process group."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _last_json(r):
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert lines, f"no JSON line; stdout: {r.stdout[-1500:]!r} stderr: {r.stderr[-1500:]!r}"
    return json.loads(lines[-1])


def _env(**extra):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    env.update({k: str(v) for k, v in extra.items()})
    return env


_WORLD1 = r"""
import json, torch
from sgam_neurips22_amd import distributed as sdist
import torch.distributed as dist
rank, local_rank, world = sdist.init(backend="nccl")
assert (rank, local_rank, world) == (0, 0, 1) and dist.is_initialized() and dist.get_backend() == "nccl"
assert sdist.collectives_active()
dev = torch.device("cuda", local_rank)
sdist.barrier()
g = sdist.gather_metrics(31, 0.125, 4242.0, dev)          # the 24-byte record all-gathered on a DEVICE tensor over RCCL
flat = torch.arange(1 << 20, dtype=torch.float32, device=dev)
want = flat.clone()
dist.all_reduce(flat)                                    # the gradient bucket's collective (world 1: identity)
dist.broadcast(flat, src=0)                              # the trainer's parameter / buffer broadcast
torch.cuda.synchronize()
ok = bool(torch.equal(flat, want))
sdist.barrier()
dist.destroy_process_group()
print(json.dumps({"g": g, "ok": ok}), flush=True)
"""


def test_nccl_world1_init_gather_teardown():
    env = _env(RANK=0, LOCAL_RANK=0, WORLD_SIZE=1, MASTER_ADDR="127.0.0.1", MASTER_PORT=_free_port())
    r = subprocess.run([sys.executable, "-c", _WORLD1], capture_output=True, text=True, env=env, timeout=280, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    out = _last_json(r)
    assert out["ok"] and out["g"]["total_frames"] == 31 and out["g"]["per_rank"] == [[31.0, 0.125, 4242.0]]
    assert abs(out["g"]["frames_per_s"] - 248.0) < 1e-9


def test_bench_under_the_launcher_command_line_world1():
    """bench.py as rank 0 of a one-rank torchrun job: RCCL process group, barrier on both sides of the timed region,
    device all-gather of the record, ONE parseable JSON line, clean exit of the launcher"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--no-secondary", "--no-roofline", "--cpu-frames", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=_env(OMP_NUM_THREADS=8), timeout=560, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    assert len(lines[0]) < 4096
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["warmup"] == 1 and out["value"] > 0
    assert out["unit"] == "frames/s" and out["scaling"] == "weak" and len(out["frame_checksums"]) == 1


_TRAIN_WORLD1 = r"""
import json, torch
from sgam_neurips22_amd import distributed as sdist, testing, training
from sgam_neurips22_amd.config import default_params
from sgam_neurips22_amd.generative_sensing_module.model import VQModel
import torch.distributed as dist
sdist.init(backend="nccl")
dev = torch.device("cuda", 0)
p = testing.small_train_params(default_params("google_earth"))
m = VQModel(**p)
m.load_state_dict(testing.synthetic_state_dict(m.state_dict(), seed=0))
m = m.to(dev).train()
tr = training.AutoencoderTrainer(m, phase="codebook", lr=1e-4)
x, mk, xd = [t.to(dev) for t in testing.train_batch()]
assert tr.synced_tensors > 0                       # the construction-time broadcast ran on device tensors over RCCL
loss0 = tr.step(x, xd, mk)
nbytes = tr.last_allreduce_bytes
loss1 = tr.step(x, xd, mk)
torch.cuda.synchronize()
dist.destroy_process_group()
print(json.dumps({"synced": tr.synced_tensors, "bucket_bytes": nbytes, "l0": float(loss0[0]), "l1": float(loss1[0])}))
"""


def test_trainer_collectives_on_device_tensors_world1():
    """the trainer's DDP half at world 1 over RCCL: parameter / buffer broadcast at construction, one flat gradient bucket
    all-reduced per update (device tensors; the world-2 gloo test of the CPU suite covers the averaging arithmetic)"""
    env = _env(RANK=0, LOCAL_RANK=0, WORLD_SIZE=1, MASTER_ADDR="127.0.0.1", MASTER_PORT=_free_port())
    r = subprocess.run([sys.executable, "-c", _TRAIN_WORLD1], capture_output=True, text=True, env=env, timeout=560, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    out = _last_json(r)
    assert out["synced"] > 100 and out["bucket_bytes"] > 0
    assert out["l0"] == out["l0"] and out["l1"] == out["l1"]          # finite losses, two updates


_SELF_LAUNCH_PROBE = r"""
import json, os, torch
from sgam_neurips22_amd import distributed as sdist
import torch.distributed as dist
rank, local_rank, world = sdist.init()
dev = torch.device("cuda", local_rank)
g = sdist.gather_metrics(7, 0.5, 11.0, dev, numa_node=sdist.pin_to_gpu_numa_node(local_rank))
sdist.barrier()
dist.destroy_process_group()
print(json.dumps({"ipc": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"), "world": world, "backend": g["rccl"]["backend"],
                  "rccl": g["rccl"], "rec": g["per_rank_records"]}), flush=True)
"""


def test_self_launch_exports_the_ipc_mode_and_the_record_carries_the_diagnostics(tmp_path):
    """`python bench.py --gpus N` without a launcher goes through distributed.self_launch: the child ranks must see
    HSA_ENABLE_IPC_MODE_LEGACY=0 even when the parent's environment does not have it (dmabuf IPC only on this host driver: RCCL's peer
    set-up fails otherwise), and the gathered record carries what a failing first multi-GPU run is diagnosed from — per-rank frames,
    seconds, checksum, NUMA node, local rank, RCCL's version and its own NCCL_DEBUG=VERSION line(s).  World 1 through the same
    launcher command line an N-GPU run uses."""
    script = tmp_path / "probe.py"
    script.write_text(_SELF_LAUNCH_PROBE)
    env = {k: v for k, v in os.environ.items() if k not in ("HSA_ENABLE_IPC_MODE_LEGACY", "NCCL_DEBUG", "NCCL_DEBUG_FILE")}
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    env["TMPDIR"] = str(tmp_path)
    code = ("import sys; from sgam_neurips22_amd import distributed as s; "
            f"sys.exit(s.self_launch({str(script)!r}, [], 1))")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=560, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    out = _last_json(r)
    assert out["ipc"] == "0" and out["world"] == 1 and out["backend"] == "nccl"
    assert out["rccl"]["version"] and out["rccl"]["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and out["rccl"]["env"]["NCCL_DEBUG"] == "VERSION"
    rec = out["rec"]
    assert len(rec) == 1 and rec[0]["rank"] == 0 and rec[0]["frames"] == 7.0 and rec[0]["checksum"] == 11.0 and rec[0]["local_rank"] == 0
    assert "numa_node" in rec[0]
