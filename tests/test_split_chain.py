"""The split-weight H chain of the marching kernel's 16-bit instances, emulated on the CPU with the constants the library
itself computes (tests/native/split_chain_check.hip; host code only -- hipcc needs no GPU for it): the hi half sums exactly,
the combined value stays inside the (0, 2 eps) window the near-integer test assumes, every unflagged sample is the reference's
(full_TB.h:58-63), and the flag rate on noise is a few in 10 000 where the single f32 chain flagged one sample in 40."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_split_chain_is_exact_where_it_claims_to_be(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "split_chain_check")
    csrc = os.path.join(ROOT, "lanczos-hls_amd", "csrc")
    subprocess.run([hipcc, "-O2", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-Wno-unused-function",
                    "-I" + os.path.join(ROOT, "include"), "-I" + csrc, os.path.join(ROOT, "tests", "native", "split_chain_check.hip"),
                    os.path.join(csrc, "lanczos_taps.cpp"), "-o", exe], check=True, timeout=600)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all cases ok" in r.stdout
    assert r.stdout.count("phase") == 3     # a = 2, 3, 4 at S = 2 (one computed phase)
