"""CPU: the oracle under AddressSanitizer + UndefinedBehaviorSanitizer (sanitizers row).
`make -C oracle asan` builds oracle/orc_selftest_asan — every translation unit of the restatement plus a driver that runs four sweeps of a
small synthetic scene through scan registration, odometry and mapping, a trust-region solve and the corner detector — and this test runs it:
exit code 0, plausible results, no sanitizer report (memory errors, leaks, signed overflow, misaligned / out-of-bounds accesses abort)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_is_clean_under_asan_and_ubsan():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    env.pop("LD_PRELOAD", None)
    r = subprocess.run([os.path.join(ROOT, "oracle", "orc_selftest_asan")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr and "LeakSanitizer" not in r.stderr, r.stderr[-4000:]
    assert "map points in the valid block" in r.stdout and "corners" in r.stdout
