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
