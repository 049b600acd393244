"""Multi-GPU plumbing of the hot path: independent sequences shard one-per-rank; the only
collective is an all-gather of the per-sequence trajectories (+ a MAX all-reduce of the elapsed time for the
bench).  Backend-agnostic: "nccl" (= RCCL over xGMI) on the GPU node, "gloo" in the CPU tests.
"""
import numpy as np


def rank_sequence_seeds(rank):
    """Seeds of the synthetic sequence rank `rank` drives (rank 0 == the single-GPU benchmark sequence)."""
    return dict(seed_scene=1234 + 17 * rank, seed_traj=42 + rank, seed_noise=5678 + 100003 * rank)


def gather_trajectories(dist, traj, max_frames, device="cpu"):
    """all_gather of fixed-size padded trajectory buffers [max_frames, 14] f64 + valid counts.

    Returns a list (one entry per rank) of [n_r, 14] arrays on every rank.  Message size <= max_frames * 112 B per
    rank (KITTI-00: 4541 frames -> 0.5 MB): latency-bound, any algorithm is fine.
    """
    import torch
    world = dist.get_world_size()
    buf = torch.zeros((max_frames, 14), dtype=torch.float64, device=device)
    n = min(traj.shape[0], max_frames)
    if n:
        buf[:n] = torch.from_numpy(np.ascontiguousarray(traj[:n])).to(device)
    cnt = torch.tensor([n], dtype=torch.int64, device=device)
    bufs = [torch.empty_like(buf) for _ in range(world)]
    cnts = [torch.empty_like(cnt) for _ in range(world)]
    dist.all_gather(bufs, buf)
    dist.all_gather(cnts, cnt)
    return [b[: int(c.item())].cpu().numpy() for b, c in zip(bufs, cnts)]


def max_over_ranks(dist, seconds, device="cpu"):
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def aggregate_throughput(steps_per_rank, world, max_seconds):
    """Whole-job scans/s: every rank processes `steps_per_rank` sweeps of its own sequence (weak scaling)."""
    return steps_per_rank * world / max_seconds


# ---------------------------------------------------------------------------------------------------------------- host side of a rank
def parse_cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11] (the format of /sys/devices/system/node/node*/cpulist)."""
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_numa_cpus(gpu_index, sysfs_root="/sys"):
    """NUMA node of the gpu_index-th GPU in KFD order (== HIP device order unless *_VISIBLE_DEVICES reorders) and that node's CPUs.

    /sys/class/kfd/kfd/topology/nodes/<n>/properties lists every HSA agent; GPUs are the nodes with simd_count > 0 and carry the DRM render
    minor, whose PCI device directory has numa_node.  Returns (node, cpus) or (None, None) when the machine does not say (no KFD, a
    single-node box reporting -1, a container without the sysfs entries)."""
    import os
    base = os.path.join(sysfs_root, "class/kfd/kfd/topology/nodes")
    try:
        nodes = sorted((int(n) for n in os.listdir(base) if n.isdigit()))
    except OSError:
        return None, None
    gpus = []
    for n in nodes:
        props = {}
        try:
            with open(os.path.join(base, str(n), "properties")) as f:
                for line in f:
                    k, _, v = line.strip().partition(" ")
                    props[k] = v
        except OSError:
            continue
        if int(props.get("simd_count", "0") or 0) > 0:
            gpus.append(int(props.get("drm_render_minor", "-1") or -1))
    if gpu_index >= len(gpus) or gpus[gpu_index] < 0:
        return None, None
    try:
        with open(os.path.join(sysfs_root, "class/drm/renderD%d/device/numa_node" % gpus[gpu_index])) as f:
            node = int(f.read().strip())
        if node < 0:
            return None, None
        with open(os.path.join(sysfs_root, "devices/system/node/node%d/cpulist" % node)) as f:
            cpus = parse_cpulist(f.read())
    except (OSError, ValueError):
        return None, None
    return (node, cpus) if cpus else (None, None)


def host_cpus_for_rank(local_rank, local_world, allowed, sysfs_root="/sys"):
    """The CPUs rank `local_rank` of `local_world` ranks on this host should run on (its enqueue thread and its ray-casting workers): the
    NUMA node of its GPU shared evenly by the ranks whose GPUs sit on the same node — or, when the topology is not available, an even
    slice of the allowed CPUs — so that eight ranks neither pile onto the same cores nor launch across the socket interconnect.
    Returns (cpus, how) with how in {"numa", "even-split"}."""
    allowed = sorted(allowed)
    node, cpus = gpu_numa_cpus(local_rank, sysfs_root)
    if node is not None:
        mine = [c for c in cpus if c in set(allowed)]
        peers = [r for r in range(local_world) if gpu_numa_cpus(r, sysfs_root)[0] == node]
        if mine and local_rank in peers:
            k, n = peers.index(local_rank), len(peers)
            share = mine[k * len(mine) // n:(k + 1) * len(mine) // n]
            if share:
                return share, "numa"
    n = max(local_world, 1)
    share = allowed[local_rank * len(allowed) // n:(local_rank + 1) * len(allowed) // n]
    return (share or allowed), "even-split"


def pin_host_threads(local_rank, local_world, sysfs_root="/sys"):
    """sched_setaffinity of this process (inherited by the workers it forks).  Returns {"cpus": n, "how": ..., "first": cpu} for the bench line."""
    import os
    if not hasattr(os, "sched_setaffinity") or local_world <= 1:
        return {"cpus": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), "how": "unpinned"}
    cpus, how = host_cpus_for_rank(local_rank, local_world, os.sched_getaffinity(0), sysfs_root)
    os.sched_setaffinity(0, cpus)
    return {"cpus": len(cpus), "how": how, "first": int(min(cpus))}


def gather_seconds(dist, seconds, device="cpu"):
    """Every rank's own elapsed time, on every rank (the bench line carries them so that a straggler is visible next to the MAX)."""
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(v.item()) for v in out]
