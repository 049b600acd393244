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


def test_rank_host_affinity_follows_the_gpu_numa_node(tmp_path):
    """multi.host_cpus_for_rank on a fake sysfs: 8 GPUs, 2 NUMA nodes of 64 CPUs (a CPU agent node in front, as KFD lists them) — every rank
    gets a quarter of ITS node's CPUs, disjoint from its neighbours'; without the topology files the allowed CPUs are split evenly."""
    import importlib
    import conftest
    conftest.load_pkg()
    multi = importlib.import_module("vloam_amd.multi")
    root = tmp_path
    nodes = root / "class/kfd/kfd/topology/nodes"
    for n in range(10):       # nodes 0, 1: the two CPU sockets; 2..9: GPUs
        d = nodes / str(n)
        d.mkdir(parents=True)
        if n < 2:
            (d / "properties").write_text("cpu_cores_count 64\nsimd_count 0\ndrm_render_minor 0\n")
        else:
            (d / "properties").write_text("cpu_cores_count 0\nsimd_count 1024\ndrm_render_minor %d\n" % (128 + n - 2))
            dev = root / ("class/drm/renderD%d/device" % (128 + n - 2))
            dev.mkdir(parents=True)
            (dev / "numa_node").write_text("%d\n" % (0 if n - 2 < 4 else 1))
    for node, lst in ((0, "0-63"), (1, "64-127")):
        d = root / ("devices/system/node/node%d" % node)
        d.mkdir(parents=True)
        (d / "cpulist").write_text(lst + "\n")
    assert multi.parse_cpulist("0-3,8,10-11") == [0, 1, 2, 3, 8, 10, 11]
    assert multi.gpu_numa_cpus(5, str(root)) == (1, list(range(64, 128)))
    allowed = set(range(128))
    shares = [multi.host_cpus_for_rank(r, 8, allowed, str(root)) for r in range(8)]
    assert all(how == "numa" for _, how in shares)
    assert [len(c) for c, _ in shares] == [16] * 8
    assert shares[0][0] == list(range(0, 16)) and shares[3][0] == list(range(48, 64)) and shares[4][0] == list(range(64, 80))
    flat = [c for cs, _ in shares for c in cs]
    assert len(flat) == len(set(flat)) == 128
    # a container that hides half of the CPUs: the share is cut from what is allowed
    half, how = multi.host_cpus_for_rank(1, 8, set(range(0, 32)), str(root))
    assert how == "numa" and half == list(range(8, 16))
    # no topology: an even split
    even = [multi.host_cpus_for_rank(r, 8, allowed, str(tmp_path / "nothing")) for r in range(8)]
    assert all(how == "even-split" for _, how in even) and [len(c) for c, _ in even] == [16] * 8
