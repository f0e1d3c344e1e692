#!/bin/bash
# usage: isa_count.sh [extra -D flags...]   (CPU) - device assembly of the library's translation unit; prints, for the K = 8 / 4-bit
# bulk kernel, the VALU / SALU / LDS / VMEM instruction counts between consecutive barriers (stage 4 is the long segment)
mkdir -p build/isa
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -Wno-pass-failed "$@" -Iinclude -Ifrisk_amd/csrc \
    --cuda-device-only -S -o build/isa/abi.s frisk_amd/csrc/frisk_abi.hip 2>/dev/null
python3 - <<'PY'
import re
L=open('/root/repo/build/isa/abi.s').read().split('\n')
for kern in ('_Z12scan8_kernelILi8ELi256ELi20ELi4ELi64ELi3ELb0ELi0EEv10ScanParams', '_Z12scan8_kernelILi8ELi256ELi20ELi8ELi64ELi2ELb0ELi0EEv10ScanParams'):
    a=next(i for i,l in enumerate(L) if l.startswith(kern+':'))
    b=next(i for i in range(a,len(L)) if 's_endpgm' in L[i])
    K=L[a:b]
    open('/root/repo/build/isa/'+('k4.s' if 'ELi4ELi64' in kern else 'k8.s'),'w').write('\n'.join(K))
    bars=[i for i,l in enumerate(K) if 's_barrier' in l]+[len(K)]
    prev=0
    print(kern[:60], 'lines', len(K))
    for bb in bars:
        seg=K[prev:bb]
        c=lambda p: sum(1 for l in seg if re.match(r'\s+'+p,l))
        print('  %5d..%5d valu %4d salu %4d lds %3d vmem %3d rcp %2d'%(prev,bb,c('v_'),c('s_'),c('ds_'),c('(global|buffer|flat|scratch)_'),c('v_rcp_f64')))
        prev=bb
PY
python3 - <<'PY'
import re
# the scoring loop's three copies (orphan list in LDS / <= 4 entries / <= 2): VALU per position in each
for f in ('k4.s','k8.s'):
    K=open('/root/repo/build/isa/'+f).read().split('\n')
    bars=[i for i,l in enumerate(K) if 's_barrier' in l]
    seg=max(zip(bars,bars[1:]), key=lambda ab: ab[1]-ab[0])
    r=[i for i in range(*seg) if 'v_rcp_f64' in K[i]]
    labels=[i for i in range(*seg) if re.match(r'\.LBB\d+_\d+:',K[i])]
    # split the reciprocal positions into runs separated by a label
    runs=[]; cur=[r[0]]
    for a,b in zip(r,r[1:]):
        if any(a<l<b for l in labels) and len(cur)>=18: runs.append(cur); cur=[b]
        else: cur.append(b)
    runs.append(cur)
    for run in runs:
        lo,hi=run[0],run[-1]
        n=len(run)-1
        c=lambda p: sum(1 for l in K[lo:hi] if re.match(r'\s+'+p,l))
        print(f,'copy at %d..%d: %d reciprocals; per position: valu %.1f salu %.1f lds %.1f vmem %.1f'%(lo,hi,len(run),c('v_')/n,c('s_')/n,c('ds_')/n,c('(global|buffer|flat|scratch)_')/n))
PY
