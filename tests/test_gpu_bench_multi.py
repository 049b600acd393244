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
