"""Scene-parallel multi-GPU support: independent trajectories shard across ranks, one process
per GPU, weights replicated, NO data-path collective.  The only communication is one tiny all-gather of the
per-rank (frames, seconds, checksum) record at the end — RCCL over xGMI on the GPU box (backend "nccl"),
gloo in the CPU tests."""
import os

# HIP streams are dealt onto hardware queues (4 by default): with fewer queues than concurrent scenes — or an unlucky
# deal — two scenes share a queue and run back to back instead of side by side (measured on one box: two scenes 319
# frames/s with the default, 447 with 8 queues; three scenes 392 vs 503).  Must be in the environment before the HIP
# runtime starts, i.e. before the first CUDA call of the process.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def env_world():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init(backend=None):
    """Initialise torch.distributed from the torchrun environment; no-op for a single process."""
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def self_launch(script, argv, n):
    """`python bench.py --gpus N` without a launcher: re-run `script argv` as N ranks of one node through
    torch.distributed.run (the same command line the driver uses), rendezvous on 127.0.0.1 and a free port.  Returns the
    launcher's exit code."""
    import socket
    import subprocess
    import sys
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this host driver (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), script] + list(argv)
    return subprocess.call(cmd, env=env)


def pin_to_gpu_numa_node(local_rank):
    """Bind this rank's host threads to the CPUs of the NUMA node its GPU hangs off (PCI bus id -> sysfs numa_node ->
    cpulist), so that the pinned staging buffer and the launch thread are local to the GPU.  Best effort: returns the
    node number, or None when the topology cannot be read (containers without sysfs, single-node hosts)."""
    try:
        prop = torch.cuda.get_device_properties(local_rank)
        bdf = f"{getattr(prop, 'pci_domain_id', 0):04x}:{prop.pci_bus_id:02x}:{prop.pci_device_id:02x}.0"
        with open(f"/sys/bus/pci/devices/{bdf}/numa_node") as f:
            node = int(f.read().strip())
        if node < 0:
            return None
        with open(f"/sys/devices/system/node/node{node}/cpulist") as f:
            cpus = set()
            for part in f.read().strip().split(","):
                lo, _, hi = part.partition("-")
                cpus.update(range(int(lo), int(hi or lo) + 1))
        if cpus:
            os.sched_setaffinity(0, cpus & os.sched_getaffinity(0) or cpus)
        return node
    except Exception:
        return None


def shard_scenes(num_scenes, rank, world):
    """rank r owns scenes {s : s mod world == r} (round-robin)."""
    return [s for s in range(num_scenes) if s % world == rank]


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def gather_metrics(frames, seconds, checksum, device):
    """All-gather one (frames, seconds, checksum) record per rank.  Returns dict(total_frames, max_seconds,
    frames_per_s, per_rank=[...]) on every rank.  24 bytes per rank: latency-bound, no bandwidth tuning."""
    rec = torch.tensor([float(frames), float(seconds), float(checksum)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        out = [torch.zeros_like(rec) for _ in range(dist.get_world_size())]
        dist.all_gather(out, rec)
    else:
        out = [rec]
    per_rank = [tuple(t.tolist()) for t in out]
    total = sum(r[0] for r in per_rank)
    tmax = max(r[1] for r in per_rank)
    return {"total_frames": total, "max_seconds": tmax, "frames_per_s": total / tmax if tmax > 0 else 0.0,
            "per_rank": per_rank}


class ConcurrentScenes:
    """Several independent trajectories on ONE GPU, each on its own HIP stream with its own model instance (captured
    graphs own static buffers, so scenes cannot share one).  A single scene is latency-bound between its ~170 dependent
    kernel launches per frame (the 16^2 / 32^2 layers fill a fraction of the chip); kernels of other scenes fill those
    gaps: aggregate frames/s rises ~1.45x with two scenes and ~1.6x with three on an MI355X, every scene still producing
    exactly the frames it produces alone (tests/test_gpu_vqgan.py).  Scenes are the path's shardable unit:
    this is the same sharding below GPU granularity; no collective, no data shared between scenes."""

    def __init__(self, make_scene, n):
        """make_scene(i) -> InfiniteSceneGeneration bound to its OWN VQModel instance (called inside stream i)."""
        self.streams = [torch.cuda.Stream() for _ in range(n)]
        self.scenes = []
        for i, s in enumerate(self.streams):
            with torch.cuda.stream(s):
                self.scenes.append(make_scene(i))

    def step(self):
        """one generated frame per scene; the host enqueues the scenes round-robin, the GPU overlaps them"""
        for sc, s in zip(self.scenes, self.streams):
            with torch.cuda.stream(s):
                sc.one_step_prediction(sc.next_pose(sc.curr))
                sc.curr += 1

    def synchronize(self):
        for s in self.streams:
            s.synchronize()
