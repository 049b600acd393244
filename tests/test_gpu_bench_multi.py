"""-m gpu: the N > 1 path of bench.py end to end — the command line the driver uses on an 8-GPU node
(python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...), here with two ranks sharing GPU 0 over the gloo
backend (VLOAM_BENCH_BACKEND=gloo; on the node the backend is "nccl" = RCCL over xGMI).  Checks what only runs with WORLD_SIZE > 1:
the per-rank sequence seeds, the barriers around the timed region, the MAX-reduced time, the all-gather of the trajectories and the
aggregate in the JSON line (BASELINE.json configs[4])."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_two_ranks_share_one_gpu_over_gloo():
    env = dict(os.environ, VLOAM_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
           "--map-warmup", "8", "--no-extras", "--no-cpu-baseline", "--synth-procs", "2"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, r.stdout[-2000:]          # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["warmup"] == 2 and d["scaling"] == "weak"
    assert d["config"]["sequences"] == 2 and d["unit"] == "scans/s"
    # whole-job aggregate: both ranks' sweeps over the slower rank's time
    assert abs(d["value"] - 2 * 6 / (d["ms_per_step"] * 6 / 1e3)) < 1e-6 * d["value"]
    g = d["config"]["gathered_trajectories"]
    assert g["ranks"] == 2 and g["frames"] == [16, 16]
    a, b = g["last_map_position"]
    assert a != b and all(abs(v) < 1e3 for v in a + b), (a, b)   # two different sequences, both gathered on rank 0
    assert d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1
    assert len(d["per_rank_ms_per_step"]) == 2 and abs(max(d["per_rank_ms_per_step"]) - d["ms_per_step"]) < 1e-9


def test_bench_eight_ranks_the_drivers_own_command_line():
    """configs[4] as the driver launches it on the 8-GPU node — `--gpus 8`, eight ranks — here sharing GPU 0 over gloo: eight different
    sequences, eight gathered trajectories, the aggregate over the slowest rank, every rank's own time in the line (a straggler shows),
    and every rank pinned to its own slice of the host's cores (NUMA node of its GPU on the node; an even split here)."""
    env = dict(os.environ, VLOAM_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "4", "--warmup", "2",
           "--map-warmup", "4", "--no-extras", "--no-cpu-baseline", "--no-kernel-timer"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["steps"] == 4 and d["scaling"] == "weak" and d["config"]["sequences"] == 8
    assert len(d["per_rank_ms_per_step"]) == 8 and abs(max(d["per_rank_ms_per_step"]) - d["ms_per_step"]) < 1e-9
    assert abs(d["value"] - 8 * 4 / (d["ms_per_step"] * 4 / 1e3)) < 1e-6 * d["value"]
    g = d["config"]["gathered_trajectories"]
    assert g["ranks"] == 8 and g["frames"] == [10] * 8
    ends = [tuple(round(v, 6) for v in p) for p in g["last_map_position"]]
    assert len(set(ends)) == 8, ends                                  # eight different sequences
    aff = d["config"]["host_affinity_rank0"]
    assert aff["how"] in ("numa", "even-split") and 1 <= aff["cpus"] <= (os.cpu_count() or 1)


def test_bench_initialises_rccl_on_one_gpu():
    """The real collective backend, once: `torchrun --nproc-per-node 1 bench.py --gpus 1` with VLOAM_BENCH_FORCE_DIST=1 takes the N > 1 path
    with backend "nccl" (= RCCL on ROCm) and world_size 1 — init_process_group with the device bound, the barriers around the timed
    region, the all_gather of the [frames, 14] f64 trajectory out of device memory and the MAX all-reduce of the elapsed time.  No scaling
    number comes out of it; it proves the RCCL import, the device binding and the environment (HSA_ENABLE_IPC_MODE_LEGACY=0, MASTER_ADDR)
    before the driver's 8-GPU run has to."""
    env = dict(os.environ, VLOAM_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("VLOAM_BENCH_BACKEND", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2",
           "--map-warmup", "8", "--no-extras", "--no-cpu-baseline", "--synth-procs", "2"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    c = d["config"]["collective"]
    assert c["backend"] == "nccl" and c["world_size"] == 1 and c["rccl_version"] and c["rccl_version"][0] >= 2, c
    g = d["config"]["gathered_trajectories"]
    assert g["ranks"] == 1 and g["frames"] == [16]
    assert d["n_gpus"] == 1 and len(d["per_rank_ms_per_step"]) == 1 and abs(d["per_rank_ms_per_step"][0] - d["ms_per_step"]) < 1e-9
