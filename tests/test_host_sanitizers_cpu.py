"""The library's host-only code under AddressSanitizer + UBSan on the CPU (the GPU pool offers no sanitizers): tools/exp/san_host.cpp
runs the 2-bit packer - the 8-letter form and, where the host has AVX-512BW + BMI2, the 64-letter form - against a letter-by-letter
restatement, bitmap -> runs, the FASTA reader and the seek index on awkward files, and the HMM on series of awkward lengths."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_headers_clean_under_asan_ubsan(tmp_path):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "san_host")
    build = subprocess.run([gxx, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-Wall", "-Wextra",
                            "-I" + os.path.join(ROOT, "frisk_amd", "csrc"), os.path.join(ROOT, "tools", "exp", "san_host.cpp"), "-o", exe,
                            "-lpthread", "-lz"], capture_output=True, text=True, timeout=600)
    if build.returncode != 0 and "sanitize" in build.stderr and "cannot find" in build.stderr:
        pytest.skip("this g++ has no sanitizer runtime")
    assert build.returncode == 0, build.stderr[-3000:]
    assert "warning" not in build.stderr, build.stderr[-3000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0 and run.stdout.strip().startswith("ok"), (run.stdout[-2000:], run.stderr[-3000:])
