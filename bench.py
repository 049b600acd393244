#!/usr/bin/env python3
"""bench.py — scans/s of the MI355X-native VLOAM per-scan odometry hot path on synthetic HDL-64E sweeps.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W``; for N > 1 launched through
``python -m torch.distributed.run --nproc-per-node N …`` (one rank per GPU, RCCL).  A *step* is one
sweep (64 rings x 2048 azimuth steps = 131 072 points) through the façade
reset -> scanRegistration -> laserOdometry [-> laserMapping] on inputs already resident in HBM.
Rank r drives its own independent sequence (different scene / trajectory / noise seeds): the path
shards one-sequence-per-GPU with no data-path collective; the only collective is an
all-gather of the trajectories after the timed region.  Rank 0 prints ONE JSON line.

PyTorch is plumbing only here (device buffers for the inputs, torch.distributed barrier / gather).
The CPU oracle (oracle/) is used only for the ``cpu_baseline`` leg and the parity numbers — never
inside the timed region.
"""
import argparse
import json
import os
import sys
import time

# Two pipelined sessions own 2 x 3 HIP streams; with the runtime's default of 4 hardware queues their stage streams share
# queues and serialise behind each other (multi-session leg: 1.0x instead of 1.75x).  Must be set before the HIP runtime loads;
# the single-session headline value does not depend on it (measured: 6 74x-6 79x scans/s either way).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s achievable

WORKLOADS = {
    "lo": "configs[1]: scanRegistration + scan-to-scan laserOdometry ICP, synthetic HDL-64E 64x2048",
    "map": "configs[2]: scanRegistration + laserOdometry + laserMapping scan-to-map ICP (voxel-hash local map), synthetic HDL-64E 64x2048",
}


def algorithmic_bytes(kernel, c):
    """ALGORITHMIC (compulsory) bytes one launch of `kernel` moves, from the measured counts of the run."""
    n_feat = c["n_sharp"] + c["n_flat"]
    F = c["F_corner"] + c["F_plane"]
    if kernel == "k_lo_assoc":  # features + both candidate clouds read once, one 76-byte factor record written per factor
        return 16 * n_feat + 16 * (c["C"] + c["S"]) + 76 * F
    if kernel == "k_lm_solve":  # every evaluation consumes the factor records (76 B each); average over the solves of a sweep
        if c["K_m"] > 0:  # 2 odometry + 2 mapping solves per sweep
            return 76 * (F * c["E_o"] + c["K_m"] * c["E_m"]) / 4.0
        return 76 * F * max(c["E_o"], 2) / 2.0
    if kernel == "k_sr_ring":   # ring-ordered cloud in, per-ring voxel centroids + picks out
        return 16 * c["N2"] + 16 * c["n_lessFlat"] + 4 * (c["n_sharp"] + c["n_lessSharp"] + c["n_flat"])
    if kernel in ("k_sr_label",):
        return 16 * c["N_in"] + 5 * c["N_in"]
    if kernel in ("k_sr_scatter",):
        return 16 * c["N_in"] + 5 * c["N_in"] + 16 * c["N2"]
    if kernel == "k_sr_compact":
        return 32 * (c["n_sharp"] + c["n_lessSharp"] + c["n_flat"] + c["n_lessFlat"])
    if kernel == "k_map_assoc":
        return (c["n_c"] + c["n_s"]) * (16 + 5 * 16) + 76 * c["K_m"]
    return 0


def pmc_traffic(workload, kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE collected in
    separate runs, FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md; tools/pmc_summary.py wrote the table).
    A PMC pass cannot run inside this process, so the number is read from profiles/ — None when the table is missing."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_%s_hbm_traffic.txt" % workload)
    tot, cnt = 0.0, 0
    try:
        for line in open(path):
            name = line.split("<")[0].split()[:1]
            f = line.split()
            if name and name[0] == kernel and len(f) >= 5:  # every instantiation of the kernel, weighted by its launches
                tot += float(f[-1]) * int(f[-4])
                cnt += int(f[-4])
    except (OSError, ValueError):
        return None, None
    if cnt == 0:
        return None, None
    return tot / cnt, os.path.relpath(path, os.path.dirname(os.path.abspath(__file__)))


def sweep_bytes(c, with_mapping):
    """B_SR + B_LO (+ B_MAP) per sweep from measured counts."""
    n_feat_all = c["n_sharp"] + c["n_lessSharp"] + c["n_flat"] + c["n_lessFlat"]
    b_sr = 16 * c["N_in"] + 16 * c["N2"] + 16 * n_feat_all
    F_o = c["F_corner"] + c["F_plane"]
    b_lo = 2 * (16 * (c["n_sharp"] + c["n_flat"]) + 16 * (c["C"] + c["S"])) + c["E_o"] * F_o * 64 + 16 * (c["n_lessSharp"] + c["n_lessFlat"])
    b_map = 0
    if with_mapping:
        nn = c["n_c"] + c["n_s"]
        b_map = 16 * (c["C"] + c["S"]) + 16 * c["M"] + 2 * nn * (16 + 5 * 16) + c["E_m"] * c["K_m"] * 64 + 16 * nn + 2 * 16 * c["M"]
    return b_sr, b_lo, b_map


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default=os.environ.get("VLOAM_BENCH_WORKLOAD", "lo"))
    ap.add_argument("--kernel", default=os.environ.get("VLOAM_BENCH_KERNEL", ""), help="kernel to bracket with HIP events (default: the dominant one)")
    ap.add_argument("--rings", type=int, default=64)
    ap.add_argument("--azimuth", type=int, default=2048)
    ap.add_argument("--cpu-sample", type=int, default=48, help="sweeps of the same sequence timed on the CPU oracle (rank 0, N=1)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU work of the cpu_baseline leg (whole passes over the sample)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timer", action="store_true", help="skip the HIP-event bracketing of the roofline kernel (for profiler runs)")
    ap.add_argument("--vo-frames", type=int, default=8, help="extra leg: frames of the VO residual stack to time (0 = skip)")
    ap.add_argument("--sessions", type=int, default=int(os.environ.get("VLOAM_BENCH_SESSIONS", "2")),
                    help="extra leg: this many independent sequences driven concurrently on ONE GPU (own handle + stream each); 0 = skip")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    K, W = args.steps, args.warmup
    with_mapping = args.workload == "map"

    import torch
    import conftest
    vl = conftest.load_pkg()
    synth = conftest.load_synth()
    import importlib
    multi = importlib.import_module("vloam_amd.multi")

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    local_rank = local_rank % torch.cuda.device_count()  # identity on a full node; lets the gloo path share one GPU in tests
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("VLOAM_BENCH_BACKEND", "nccl")  # "nccl" == RCCL; "gloo" only to exercise this path on a 1-GPU box
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)

    coll_dev = "cuda" if os.environ.get("VLOAM_BENCH_BACKEND", "nccl") == "nccl" else "cpu"
    # ---- synthetic input, one independent sequence per rank, resident in HBM before the timed region
    seq = synth.SynthSequence(n_rings=args.rings, n_azimuth=args.azimuth, n_sweeps=W + K, **multi.rank_sequence_seeds(rank))
    host = np.stack([seq.sweep(k) for k in range(W + K)])
    n_pts = host.shape[1]
    d_clouds = torch.from_numpy(host).to(torch.device("cuda", local_rank))
    base_ptr, stride = d_clouds.data_ptr(), n_pts * 16

    h = vl.Handle(local_rank, scan_line=args.rings, with_mapping=int(with_mapping), max_points=max(n_pts, 1024), max_frames=W + K + 8)
    # the kernel with the largest share of GPU time in the committed rocprofv3 summaries (profiles/r01_{lo,map}_kernel_stats.txt)
    kernel = args.kernel or ("k_lm_solve" if with_mapping else "k_sr_ring")

    def barrier():
        if dist is not None:
            dist.barrier()

    for k in range(W):
        h.process_scan_device(base_ptr + k * stride, n_pts)
    h.sync()

    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(W, W + K):
        h.process_scan_device(base_ptr + k * stride, n_pts)
    h.sync()
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()

    elapsed = t1 - t0
    if dist is not None:
        elapsed = multi.max_over_ranks(dist, elapsed, device=coll_dev)
    counts = h.counts()
    traj = h.trajectory()

    # ---- roofline leg: the dominant kernel's launch duration, HIP events on the kernel's own stream.  A separate replay of
    # the same K sweeps on a fresh session, so that the event records (marker packets around every launch of that kernel)
    # do not sit in the timed region above.
    hk = vl.Handle(local_rank, scan_line=args.rings, with_mapping=int(with_mapping), max_points=max(n_pts, 1024), max_frames=W + K + 8)
    for k in range(W):
        hk.process_scan_device(base_ptr + k * stride, n_pts)
    hk.sync()
    if not args.no_kernel_timer:
        hk.profile_kernel(kernel, 8 * K + 16)
    for k in range(W, W + K):
        hk.process_scan_device(base_ptr + k * stride, n_pts)
    hk.sync()
    k_ms, k_launches = hk.profile_read() if not args.no_kernel_timer else (0.0, 0)
    hk.close()

    # the one collective of the path: gather the per-sequence trajectories
    trajectories = [traj]
    if dist is not None:
        trajectories = multi.gather_trajectories(dist, traj, W + K + 8, device=coll_dev)

    # ---- extra leg (reported separately, never the headline): multi-session throughput of one GPU.  A single sequence is a
    # chain of dependent launches (latency bound); independent sessions on separate streams overlap those latencies.
    multi_session = None
    if args.sessions > 1 and world == 1:
        import threading
        B = args.sessions
        hs = [vl.Handle(local_rank, scan_line=args.rings, with_mapping=int(with_mapping), max_points=max(n_pts, 1024), max_frames=W + K + 8)
              for _ in range(B)]

        def drive(hh, lo, hi):
            for kk in range(lo, hi):
                hh.process_scan_device(base_ptr + kk * stride, n_pts)
            hh.sync()

        ths = [threading.Thread(target=drive, args=(hh, 0, W)) for hh in hs]
        [t.start() for t in ths]
        [t.join() for t in ths]
        torch.cuda.synchronize()
        m0 = time.perf_counter()
        ths = [threading.Thread(target=drive, args=(hh, W, W + K)) for hh in hs]
        [t.start() for t in ths]
        [t.join() for t in ths]
        torch.cuda.synchronize()
        m1 = time.perf_counter()
        same = all(np.array_equal(hh.trajectory(), traj) for hh in hs)
        multi_session = {"sessions": B, "value": B * K / (m1 - m0), "unit": "scans/s", "ms_per_step_all_sessions": 1e3 * (m1 - m0) / K,
                         "trajectories_identical_to_single_session": bool(same),
                         "note": "B independent handles (3 streams each, one host thread each) on one GPU replaying the same sweeps; one pipelined session "
                                 "already keeps the launch path busy; beyond ~1.8x the sessions are host-launch bound (all launches of one process go "
                                 "through the runtime's queue locks); GPU_MAX_HW_QUEUES=" + os.environ.get("GPU_MAX_HW_QUEUES", "") + "; not the headline value"}
        for hh in hs:
            hh.close()

    # ---- extra: latency of ONE sweep (enqueue + drain, nothing in flight), on a fresh session.  The headline value streams the
    # sequence: the three stage streams overlap consecutive sweeps, so 1 / value is a throughput period, not a latency.
    latency = None
    if world == 1:
        hl = vl.Handle(local_rank, scan_line=args.rings, with_mapping=int(with_mapping), max_points=max(n_pts, 1024), max_frames=W + 40)
        for k in range(W):
            hl.process_scan_device(base_ptr + k * stride, n_pts)
        hl.sync()
        L = min(32, K)
        l0 = time.perf_counter()
        for k in range(W, W + L):
            hl.process_scan_device(base_ptr + k * stride, n_pts)
            hl.sync()
        l1 = time.perf_counter()
        latency = {"ms_per_sweep": 1e3 * (l1 - l0) / L, "sweeps": L,
                   "note": "one sweep at a time (vloam_sync after each): no overlap between consecutive sweeps"}
        hl.close()

    # ---- extra: the depth-enhanced VO residual stack (configs[3]'s GPU part) on synthetic matches, host-synchronous API
    # (cloud and matches come from host memory, the solved motion goes back): vloam_vo_process_point_cloud + vloam_vo_solve
    vo_stage = None
    if world == 1 and args.vo_frames > 0:
        hv = vl.Handle(local_rank, scan_line=args.rings, with_mapping=0, max_points=max(n_pts, 1024))
        hv.vo_set_calib(*synth.kitti_like_calib())
        nf = min(args.vo_frames, W + K - 1)
        ms_ = [synth.synth_matches(seq, k) for k in range(1, nf + 1)]
        hv.vo_process_point_cloud(host[0])
        v0 = time.perf_counter()
        for k in range(1, nf + 1):
            hv.vo_process_point_cloud(host[k])
            hv.vo_solve(ms_[k - 1][0], ms_[k - 1][1], np.zeros(3), np.zeros(3))
        v1 = time.perf_counter()
        vo_stage = {"ms_per_frame": 1e3 * (v1 - v0) / nf, "frames": nf, "matches": int(ms_[0][0].shape[0]),
                    "note": "projection + 5-px bucket depth map + 3-NN depth lookup + K^-1 QR + <=100-iteration LM, incl. the 2 MB H2D copy"}
        hv.close()

    out = None
    if rank == 0:
        value = multi.aggregate_throughput(K, world, elapsed)
        assert len(trajectories) == world and all(t.shape == (W + K, 14) for t in trajectories)
        b_sr, b_lo, b_map = sweep_bytes(counts, with_mapping)
        kb = algorithmic_bytes(kernel, counts)
        avg_ms = k_ms / max(k_launches, 1)
        traffic, traffic_src = pmc_traffic(args.workload, kernel)
        achieved = kb / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        out = {
            "metric": "scans/sec end-to-end odometry on 64x2048 cloud", "value": value, "unit": "scans/s", "n_gpus": world, "steps": K,
            "warmup": W, "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 points / f64 poses+residuals", "data": "synthetic",
            "config": {"workload": WORKLOADS[args.workload], "points_per_sweep": int(n_pts), "sequences": world,
                       "sharding": "one independent sequence per GPU, no data-path collective; all_gather of trajectories after the run"},
            "roofline": {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": kb,
                         "avg_launch_us": 1e3 * avg_ms, "launches_timed": k_launches,
                         "sweep_algorithmic_bytes": {"B_SR": b_sr, "B_LO": b_lo, "B_MAP": b_map},
                         "end_to_end_frac": (b_sr + b_lo + b_map) * value / world / 1e9 / HBM_PEAK_GBS},
            "counts_last_sweep": counts,
        }
        out["config"]["pipelining"] = "SR / LO / mapping of consecutive sweeps overlap on three HIP streams (one sequence, one GPU)"
        if latency:
            out["latency"] = latency
        if vo_stage:
            out["vo_stage"] = vo_stage
        if multi_session:
            out["multi_session"] = multi_session
        if world == 1 and not args.no_cpu_baseline:
            import orc
            ns = min(args.cpu_sample, W + K)
            done, c0 = 0, time.perf_counter()
            while True:  # whole passes over the first ns sweeps (fresh oracle session each) until ~10 s of CPU work are on the clock
                o = orc.Oracle(scan_line=args.rings, with_mapping=with_mapping)
                for k in range(ns):
                    o.process(host[k])
                done += ns
                if time.perf_counter() - c0 >= args.cpu_seconds:
                    break
            c1 = time.perf_counter()
            out["cpu_baseline"] = {"value": done / (c1 - c0), "unit": "scans/s", "cores": 1, "kind": "port",
                                   "sample": "%d sweeps (%d passes over the first %d sweeps of the same synthetic sequence, %.1f s) through "
                                             "the CPU oracle (restated reference path, single thread like the reference)"
                                             % (done, done // ns, ns, c1 - c0)}
            # parity of the sample: re-run the same sweeps on a fresh handle and compare poses frame by frame
            hp = vl.Handle(local_rank, scan_line=args.rings, with_mapping=int(with_mapping), max_points=max(n_pts, 1024))
            dt_max = dq_max = 0.0
            np_ = min(ns, 12)
            for k in range(np_):
                hp.process_scan(host[k])
            hp.sync()
            tj = hp.trajectory()
            o3 = orc.Oracle(scan_line=args.rings, with_mapping=with_mapping)
            for k in range(np_):
                o3.process(host[k])
                qw, tw, _, _ = o3.lo_pose()
                dt_max = max(dt_max, float(np.linalg.norm(tj[k, 4:7] - tw)))
                dq_max = max(dq_max, float(min(np.linalg.norm(tj[k, 0:4] - qw), np.linalg.norm(tj[k, 0:4] + qw))))
                if with_mapping:
                    qm, tm, _, _ = o3.map_pose()
                    dt_max = max(dt_max, float(np.linalg.norm(tj[k, 11:14] - tm)))
                    dq_max = max(dq_max, float(min(np.linalg.norm(tj[k, 7:11] - qm), np.linalg.norm(tj[k, 7:11] + qm))))
            out["parity_vs_oracle"] = {"frames": np_, "max_abs_dt_m": dt_max, "max_abs_dq": dq_max, "bar": 1e-4}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
