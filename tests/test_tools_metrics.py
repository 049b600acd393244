"""tools/run_sequence.py --metrics: the per-frame count of points whose azimuth sits on one of scanRegistration's unwrap thresholds
(scan_registration.cpp:236-262) — the points whose relTime may differ by a revolution between math libraries.  Host-side numpy only."""
import importlib.util
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    sp = importlib.util.spec_from_file_location("run_sequence_tool", os.path.join(ROOT, "tools", "run_sequence.py"))
    m = importlib.util.module_from_spec(sp)
    sp.loader.exec_module(m)
    return m


def _ring(az, r=30.0):
    c = np.zeros((az.size, 4), dtype=np.float32)
    c[:, 0], c[:, 1] = r * np.cos(az), r * np.sin(az)
    return c


def test_points_on_an_unwrap_threshold_are_counted():
    m = _tool()
    az = np.linspace(0.1, 0.1 - 2 * np.pi * 1.001, 4000)        # one clockwise revolution, like a spinning lidar
    cloud = _ring(az)
    assert m.unwrap_boundary_points(cloud) == 0                 # thresholds fall between samples
    # a return exactly half a revolution after the first one sits on the halfPassed threshold (ori - startOri == pi)
    start = -np.arctan2(cloud[0, 1].astype(np.float64), cloud[0, 0].astype(np.float64))
    hit = _ring(np.array([-(start + np.pi)]))
    both = np.vstack([cloud[:2000], hit, cloud[2000:]])
    assert m.unwrap_boundary_points(both) >= 1
    assert m.unwrap_boundary_points(both, band=1e-12) <= m.unwrap_boundary_points(both)
    # NaN / closer-than-minimum-range points never reach the azimuth test
    junk = np.full((10, 4), np.nan, dtype=np.float32)
    assert m.unwrap_boundary_points(np.vstack([junk, cloud])) == 0
    assert m.unwrap_boundary_points(junk) == 0


def test_bench_refuses_a_traffic_table_of_other_kernels(tmp_path, monkeypatch):
    """bench.py quotes roofline.traffic from the committed PMC table only if the table's csrc hash is the hash of the kernels it times."""
    import importlib.util
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import pmc_summary
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    csrc = tmp_path / "vloam-cmu-16833_amd" / "csrc"
    csrc.mkdir(parents=True)
    (csrc / "a.hip").write_text("__global__ void k() {}\n")
    (tmp_path / "profiles").mkdir()
    sha = pmc_summary.csrc_sha256(str(tmp_path))
    table = "# HBM traffic\n# csrc_sha256: %s\nkernel launches fetch write total\nk_lm_solve<true, 2, 6>   10   100   50   150\nk_lm_solve<true, 1, 4>   30   10   40   50\n"
    (tmp_path / "profiles" / "r09_map_hbm_traffic.txt").write_text(table % sha)
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(pmc_summary, "csrc_sha256", lambda root=None, _f=pmc_summary.csrc_sha256: _f(str(tmp_path)))
    val, src = bench.pmc_traffic("map", "k_lm_solve")
    assert abs(val - (150 * 10 + 50 * 30) / 40.0) < 1e-9 and src.endswith("r09_map_hbm_traffic.txt")
    (csrc / "a.hip").write_text("__global__ void k() { }\n")          # the kernels changed after the PMC pass
    val, src = bench.pmc_traffic("map", "k_lm_solve")
    assert val is None and "STALE" in src
    (tmp_path / "profiles" / "r09_map_hbm_traffic.txt").write_text(table.replace("# csrc_sha256: %s\n", ""))   # an old table without a hash
    val, src = bench.pmc_traffic("map", "k_lm_solve")
    assert val is None and "STALE" in src
