"""CPU sanitizer runs ("compile CPU oracle with -fsanitize=address,undefined"; GPU AddressSanitizer is not available on
this pool).  Two instrumented executables, built here with gcc / g++ and run once each:
  * the product's host side — vqt_host.cpp, analysis_host.cpp, consumers_host.cpp, multi_host.cpp — behind tests/sanitize/host_main.cpp,
    which replays the inputs of test_host_plan / test_analysis_state / test_consumers / test_multi_device;
  * the oracle, oracle/pvq_oracle.c, behind tests/sanitize/oracle_main.c.
Any heap / stack overrun, use after free, signed overflow, misaligned or out-of-range access aborts the run (ASan exits non-zero,
UBSan runs with -fno-sanitize-recover)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-g", "-O1", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not installed")
def test_host_side_under_asan_ubsan(tmp_path):
    csrc = os.path.join(ROOT, "pitchvis_amd", "csrc")
    exe = str(tmp_path / "host_san")
    srcs = [os.path.join(csrc, f) for f in ("vqt_host.cpp", "analysis_host.cpp", "consumers_host.cpp", "multi_host.cpp")]
    cmd = ["g++", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-Wall", *SAN, "-I", csrc,
           os.path.join(ROOT, "tests", "sanitize", "host_main.cpp"), *srcs, "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=ENV)
    assert r.returncode == 0 and "SANITIZE_HOST_OK" in r.stdout, (r.stdout[-1000:], r.stderr[-4000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not installed")
def test_oracle_under_asan_ubsan(tmp_path):
    orc = os.path.join(ROOT, "oracle")
    exe = str(tmp_path / "oracle_san")
    cmd = ["gcc", "-std=gnu11", "-ffp-contract=off", "-fno-fast-math", "-Wall", *SAN, "-I", orc,
           os.path.join(ROOT, "tests", "sanitize", "oracle_main.c"), os.path.join(orc, "pvq_oracle.c"), "-o", exe, "-lm"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900, env=ENV)
    assert r.returncode == 0 and "SANITIZE_ORACLE_OK" in r.stdout, (r.stdout[-1000:], r.stderr[-4000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]
