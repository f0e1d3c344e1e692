#!/usr/bin/env python3
"""Per-kernel resource table of the built library (CPU): registers, spills, scratch and LDS of every kernel in
frisk_amd/libfrisk_hip.so, read from the code object's metadata notes (clang-offload-bundler + llvm-readelf --notes).
Usage: python tools/kernel_resources.py [lib.so] [> profiles/rN_kernel_resources.txt]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "frisk_amd", "libfrisk_hip.so")
    tmp = tempfile.mkdtemp(prefix="frisk_res_")
    fat = os.path.join(tmp, "fat.bin")
    subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
    co = os.path.join(tmp, "gfx950.co")
    subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True)
    notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], check=True, capture_output=True, text=True).stdout
    demangle = lambda n: subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()  # noqa: E731
    rows = []
    for block in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
        get = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, block) or [None, "?"])[1]      # noqa: E731
        name = get("name")
        rows.append((demangle(name).replace("(ScanParams)", ""), get("vgpr_count"), get("vgpr_spill_count"), get("sgpr_count"),
                     get("sgpr_spill_count"), get("private_segment_fixed_size"), get("group_segment_fixed_size")))
    print("%-70s %5s %6s %5s %6s %8s %7s" % ("kernel", "vgpr", "vspill", "sgpr", "sspill", "scratchB", "ldsB"))
    for r in sorted(rows):
        print("%-70s %5s %6s %5s %6s %8s %7s" % r)


if __name__ == "__main__":
    main()
