#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as the guide prescribes).

Units / corrections (MI355X_MICROARCH.md §HBM): both counters are in KiB; on gfx950 FETCH_SIZE tallies the 128-B requests of a
wide coalesced stream at 64 B, i.e. reports HALF the bytes — doubled here (an upper bound for narrow / scattered accesses, where
the calibration is unknown).  WRITE_SIZE is taken as is (uncalibrated).  Output: average bytes per launch.
"""
import csv
import collections
import glob
import hashlib
import os
import sys


def csrc_sha256(root=None):
    """Content hash of the kernel sources (csrc/*.hip, *.h, *.cpp, file names included): written into every PMC table so that bench.py can
    refuse to quote a traffic figure measured on other kernels than the ones it is timing (there is no .git on the GPU box)."""
    root = root or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(root, "vloam-cmu-16833_amd", "csrc", "*"))):
        if os.path.isfile(path) and path.endswith((".hip", ".h", ".cpp")):
            h.update(os.path.basename(path).encode())
            h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


def load(path, counter):
    tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] != counter:
            continue
        name = row["Kernel_Name"].split("(")[0].replace("vloam::", "").replace("void ", "")
        tot[name] += float(row["Counter_Value"])
        cnt[name] += 1
    return tot, cnt


def main(fetch_csv, write_csv, out_path):
    ft, fc = load(fetch_csv, "FETCH_SIZE")
    wt, wc = load(write_csv, "WRITE_SIZE")
    lines = ["# HBM traffic per launch from rocprofv3 --pmc (FETCH_SIZE x2 gfx950 correction, WRITE_SIZE raw), bytes",
             "# csrc_sha256: %s" % csrc_sha256(),
             "%-28s %8s %16s %16s %16s" % ("kernel", "launches", "fetch_B/launch", "write_B/launch", "total_B/launch")]
    for k in sorted(ft, key=lambda k: -(ft[k] + wt.get(k, 0))):
        f = 2.0 * 1024.0 * ft[k] / max(fc[k], 1)
        w = 1024.0 * wt.get(k, 0.0) / max(wc.get(k, 1), 1)
        lines.append("%-28s %8d %16.0f %16.0f %16.0f" % (k, fc[k], f, w, f + w))
    open(out_path, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main(*sys.argv[1:4])
