#!/bin/bash
# usage: build_tune.sh <out.so> [extra -D flags...]
out=$1; shift
cd /root/repo
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -shared -fPIC -ffp-contract=off -Wno-pass-failed -DFRISK_TUNE "$@" -Iinclude -Ifrisk_amd/csrc -Rpass-analysis=kernel-resource-usage -o $out frisk_amd/csrc/frisk_abi.hip -lz 2> ${out%.so}.res.txt
rc=$?
grep -E "error" ${out%.so}.res.txt | head
python3 - "${out%.so}.res.txt" <<'PY'
import re,sys
txt=open(sys.argv[1]).read()
blocks=re.split(r"remark: [^\n]*Function Name: ", txt)
for b in blocks[1:]:
    name=b.split("\n")[0]
    if 'scan8' not in name and 'scan_kernel' not in name: continue
    g=lambda k: re.search(k+r": (\d+)", b)
    vals=[(g(k).group(1) if g(k) else '?') for k in ["VGPRs","SGPRs","ScratchSize \[bytes/lane\]","Occupancy \[waves/SIMD\]"]]
    print(name[:60], "vgpr/sgpr/scratch/occ", vals)
PY
exit $rc
