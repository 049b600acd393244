"""CPU: the N > 1 path (one independent sequence per rank, all_gather of trajectories, MAX-reduced time) on the gloo
backend with world_size 2.  The per-rank odometry here is produced by the CPU oracle (test infrastructure) — the
point of the test is the sharding / collective plumbing that bench.py runs over RCCL."""
import os
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import importlib
    import torch.distributed as dist
    import conftest
    import orc
    conftest.load_pkg()
    multi = importlib.import_module("vloam_amd.multi")
    synth = conftest.load_synth()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    seq = synth.SynthSequence(n_rings=64, n_azimuth=256, n_sweeps=3, **multi.rank_sequence_seeds(rank))
    o = orc.Oracle(with_mapping=False)
    rows = []
    for k in range(3 if rank == 0 else 2):  # ragged: ranks may hold different frame counts
        o.process(seq.sweep(k))
        qw, tw, _, _ = o.lo_pose()
        rows.append(np.concatenate([qw, tw, qw, tw]))
    traj = np.array(rows)
    gathered = multi.gather_trajectories(dist, traj, max_frames=8)
    tmax = multi.max_over_ranks(dist, 1.0 + rank)
    np.savez(os.path.join(out_dir, "r%d.npz" % rank), tmax=tmax, n=[g.shape[0] for g in gathered], own=traj,
             g0=gathered[0], g1=gathered[1])
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather(tmp_path):
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "r0.npz"), np.load(tmp_path / "r1.npz")
    assert list(r0["n"]) == [3, 2] and list(r1["n"]) == [3, 2]
    assert r0["tmax"] == 2.0 and r1["tmax"] == 2.0
    assert np.array_equal(r0["g0"], r0["own"]) and np.array_equal(r0["g1"], r1["own"])   # every rank sees every trajectory
    assert np.array_equal(r1["g0"], r0["own"])
    assert not np.allclose(r0["own"][2, 4:7], r1["own"][1, 4:7])                          # different seeds -> different motion


def test_aggregate_and_seeds():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import importlib
    import conftest
    conftest.load_pkg()
    multi = importlib.import_module("vloam_amd.multi")
    assert multi.aggregate_throughput(64, 8, 0.5) == 64 * 8 / 0.5
    assert multi.rank_sequence_seeds(0) == dict(seed_scene=1234, seed_traj=42, seed_noise=5678)   # SURVEY.md §8d seeds
    assert len({tuple(multi.rank_sequence_seeds(r).values()) for r in range(8)}) == 8
