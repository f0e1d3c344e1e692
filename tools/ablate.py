#!/usr/bin/env python3
"""Kernel experiments: build variants of libfrisk_hip.so with -DFRISK_ABL=<mask> (the ingredient mask scan_kernel.h tests; or other -D flags) and time
the scan / profile kernels of each on one synthetic shard.  Results of ablated builds are WRONG by design; only
the timings mean anything.  Usage (on the GPU box): python tools/ablate.py 0 1 2 4 8 15"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "frisk_amd", "csrc")
OUT = os.path.join(ROOT, "gpurun_out", "ablate")

CHILD_K = int(os.environ.get("ABLATE_K", "8"))
CHILD = r'''
import sys, json
sys.path.insert(0, %r)
from frisk_amd import Engine, synth
lens = [int(x*%f) for x in synth.c5_shard_lens(8, 0)]
e = Engine(1, %d)
e.synth(lens, seed=0xC5, island_frac=0.02, n_frac=0.07, lower_frac=%f, repeats_per_kb=%f)
e.profile_reset(); e.profile_add(); e.profile_finalize()
ts = []
for _ in range(4):
    r = e.scan(5000, 1000, pinned=True); ts.append(e.kernel_ms(0))
e.profile_reset(); e.profile_add(); tp = e.kernel_ms(1)
print(json.dumps({"scan_ms": min(ts), "profile_ms": tp, "cands": r.n_candidates, "kld_sum": float(r.kld[r.kept].sum()), "stat": e.scan_stat()}))
'''


def main():
    os.makedirs(OUT, exist_ok=True)
    scale = float(os.environ.get("ABLATE_SCALE", "0.25"))
    for spec in sys.argv[1:]:
        defs = ["-DFRISK_ABL=" + spec, "-DFRISK_TUNE", "-DFRISK_K8_WIDTH=16"] if spec.isdigit() else ["-D" + d for d in spec.split(",")]
        lib = os.path.join(OUT, "lib_%s.so" % spec.replace(",", "_").replace("=", ""))
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC",
                        "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-o", lib,
                        os.path.join(CSRC, "frisk_abi.hip"), "-lz"] + defs, check=True)
        env = dict(os.environ, FRISK_HIP_LIB=lib)
        out = subprocess.run([sys.executable, "-c", CHILD % (ROOT, scale, CHILD_K, float(os.environ.get("ABLATE_LOWER", "0")), float(os.environ.get("ABLATE_REPEATS", "0")))], env=env, capture_output=True, text=True)
        print(spec, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-500:], flush=True)
        for line in out.stderr.splitlines():
            if line.startswith("[stamps]"):
                print("   ", line, flush=True)
                break


if __name__ == "__main__":
    main()
