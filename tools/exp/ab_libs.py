#!/usr/bin/env python3
"""(GPU) same-box A/B of two builds of the library: usage ab_libs.py <srcdirA> <srcdirB> [rounds] - each dir holds frisk_amd/csrc and include;
builds both, then times the bench shard's scan (HIP events, best of 5) with each in turn, `rounds` times."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "gpurun_out", "ablate"); os.makedirs(OUT, exist_ok=True)
CHILD = r'''
import sys, json
sys.path.insert(0, %r)
from frisk_amd import Engine, synth
lens = synth.c5_shard_lens(8, 0)
e = Engine(1, 8)
e.synth(lens, seed=0xC5, island_frac=0.02, n_frac=0.07, lower_frac=0.0)
e.profile_reset(); e.profile_add(); e.profile_finalize()
ts = []
for _ in range(8):
    r = e.scan(5000, 1000, pinned=True); ts.append(e.kernel_ms(0))
print(json.dumps({"scan_ms_best": round(min(ts), 4), "scan_ms_last3": [round(t, 3) for t in ts[-3:]], "kld_sum": float(r.kld[r.kept].sum())}))
''' % ROOT
libs = []
for i, d in enumerate(sys.argv[1:3]):
    lib = os.path.join(OUT, "lib_ab%d.so" % i)
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off", "-Wno-pass-failed",
                    "-I" + os.path.join(d, "include"), "-I" + os.path.join(d, "frisk_amd", "csrc"), "-o", lib,
                    os.path.join(d, "frisk_amd", "csrc", "frisk_abi.hip"), "-lz"], check=True, stderr=subprocess.DEVNULL)
    libs.append(lib)
for rnd in range(int(sys.argv[3]) if len(sys.argv) > 3 else 3):
    for name, lib in zip("AB", libs):
        out = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, FRISK_HIP_LIB=lib), capture_output=True, text=True)
        print(name, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:], flush=True)
