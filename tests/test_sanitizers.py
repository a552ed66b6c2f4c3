"""the CPU-side C under -fsanitize=address,undefined (GPU sanitizers are not available on this pool).

  * oracle/ (the restatement of full_TB.h:29-96) + oracle/selftest.c: every loop bound clipped, u8/u16, threads
  * lanczos-hls_amd/host/image_io.c + image_io_selftest.c: PNG/PNM round trips, hostile headers
A sanitizer report aborts the driver (-fno-sanitize-recover=all) and fails the test; the oracle's digests must also equal
those of the ordinary -O2 build.
"""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")


def test_oracle_under_asan_ubsan():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan", "--no-print-directory"], check=True,
                   stdout=subprocess.DEVNULL)
    san = subprocess.run([os.path.join(ROOT, "oracle", "_san", "oracle_selftest")], capture_output=True, text=True, env=ENV)
    assert san.returncode == 0, san.stderr[-2000:]
    plain = subprocess.run([os.path.join(ROOT, "oracle", "_san", "oracle_selftest_plain")], capture_output=True, text=True)
    assert plain.returncode == 0
    assert san.stdout == plain.stdout and san.stdout.count("\n") >= 60


def test_host_image_io_under_asan_ubsan(tmp_path):
    subprocess.run(["make", "-C", os.path.join(ROOT, "lanczos-hls_amd"), "san", "--no-print-directory"], check=True,
                   stdout=subprocess.DEVNULL)
    r = subprocess.run([os.path.join(ROOT, "lanczos-hls_amd", "build_san", "image_io_selftest"), str(tmp_path)],
                       capture_output=True, text=True, env=ENV)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


def test_cache_lifetime_under_asan_ubsan(tmp_path):
    """create -> many shapes -> destroy of the host-only parts (table cache, retire lists, plan-cache recency order) against a model of
    streams in which queued copies complete only when the test says so (tests/native/cache_lifetime_check.cpp; round-3 harness
    crash)."""
    exe = str(tmp_path / "cache_lifetime_check")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-Wall", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                    "-I" + os.path.join(ROOT, "lanczos-hls_amd", "csrc"), os.path.join(ROOT, "tests", "native", "cache_lifetime_check.cpp"),
                    "-o", exe], check=True, timeout=300)
    r = subprocess.run([exe], capture_output=True, text=True, env=ENV, timeout=120)
    assert r.returncode == 0 and "cache lifetime: all cases ok" in r.stdout, r.stdout + r.stderr[-2000:]
