"""csrc/fdlibm_f32.h (the atanf / atan2f the device's scan registration calls) against the C library's atanf / atan2f, bit for bit.

The reference computes a return's scan line and relative time from glibc's float atan / atan2 (scan_registration.cpp:166-167,192,234); the
header restates glibc's (fdlibm's) algorithm so that the device produces the same bits.  Compiled here for the host with g++
-ffp-contract=off (the flag the device translation unit is built with) and run over 2^24 arguments per generator: random bit patterns of
every binade, the arguments LiDAR returns produce, break points, signed zeros, infinities, NaN.  No GPU involved; the GPU side of the same
claim is tests/test_gpu_scan_registration.py (intensity and startOri / endOri bit for bit against the oracle, which calls the C library).
The check is meaningful on a glibc that still builds atanf / atan2f from the fdlibm float sources (<= 2.40; this image: 2.35)."""
import os
import platform
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_restated_atanf_atan2f_equal_the_c_library(tmp_path):
    libc = platform.libc_ver()
    if libc[0] == "glibc" and tuple(int(v) for v in libc[1].split(".")[:2]) >= (2, 41):
        pytest.skip("glibc %s computes atanf / atan2f with the correctly-rounded CORE-MATH routines, not fdlibm's" % libc[1])
    exe = str(tmp_path / "fdlibm_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(ROOT, "vloam-cmu-16833_amd", "csrc"),
                    os.path.join(ROOT, "tests", "cpp", "fdlibm_check.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe, str(1 << 24)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:]
    rows = dict((ln.split()[0], [int(v) for v in ln.split()[1:]]) for ln in r.stdout.strip().splitlines()[-2:])
    assert rows["atanf"][0] > (1 << 25) and rows["atanf"][1] == 0
    assert rows["atan2f"][0] > (1 << 25) and rows["atan2f"][1] == 0
