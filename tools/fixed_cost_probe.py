import os, sys, time
import numpy as np
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest
vl = conftest.load_pkg(); synth = conftest.load_synth()
import torch
for n_az in (2048, 512, 128):
    n = 240
    seq = synth.SynthSequence(n_rings=64, n_azimuth=n_az, n_sweeps=48)
    host = np.stack([seq.sweep(k) for k in range(48)])
    d = torch.from_numpy(host).cuda()
    npts = host.shape[1]
    for mapping in (1, 0):
        h = vl.Handle(0, with_mapping=mapping, max_points=max(npts, 1024), max_frames=n + 8)
        for k in range(40): h.process_scan_device(d.data_ptr() + (k % 48) * npts * 16, npts)
        h.sync()
        t0 = time.perf_counter(); tc = 0.0
        for k in range(40, n):
            c0 = time.perf_counter()
            h.process_scan_device(d.data_ptr() + (k % 48) * npts * 16, npts)
            tc += time.perf_counter() - c0
        t1 = time.perf_counter()
        h.sync()
        t2 = time.perf_counter()
        print("n_az %4d mapping %d: %.0f scans/s, period %.1f us, host time inside the calls %.1f us per sweep (enqueue loop %.1f us per sweep)" %
              (n_az, mapping, (n - 40) / (t2 - t0), 1e6 * (t2 - t0) / (n - 40), 1e6 * tc / (n - 40), 1e6 * (t1 - t0) / (n - 40)))
        h.close()
