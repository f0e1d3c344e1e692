#!/bin/bash
# usage: build_variant.sh <name> [-D flags...]  ->  build/ab/<name>.so (+ register / scratch report of the scan kernels in build/ab/<name>.res.txt)
name=$1; shift
mkdir -p build/ab
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -shared -fPIC -ffp-contract=off -Wno-pass-failed "$@" -Iinclude -Ifrisk_amd/csrc -Rpass-analysis=kernel-resource-usage -o build/ab/$name.so frisk_amd/csrc/frisk_abi.hip -lz 2> build/ab/$name.res.txt
rc=$?
grep -E "error" -A3 build/ab/$name.res.txt | head -20
exit $rc
