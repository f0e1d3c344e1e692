#!/usr/bin/env python3
"""(CPU) device assembly of the library; for the K = 8 / 4-bit bulk kernels (plain and SIDE) the copies of the scoring loop:
instructions per position and scratch accesses in each.  Usage: python tools/exp/isa_copies.py [extra -D flags]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.makedirs(os.path.join(ROOT, "build/isa"), exist_ok=True)
asm = os.path.join(ROOT, "build/isa/abi.s")
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-pass-failed"] + sys.argv[1:] +
               ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "frisk_amd/csrc"), "--cuda-device-only", "-S", "-o", asm,
                os.path.join(ROOT, "frisk_amd/csrc/frisk_abi.hip")], check=True, stderr=subprocess.DEVNULL)
L = open(asm).read().split('\n')
for tag, kern in (('plain', '_Z12scan8_kernelILi8ELi256ELi20ELi4ELi64ELi3ELb0ELi0ELb0EEv10ScanParams'),
                  ('side', '_Z12scan8_kernelILi8ELi256ELi20ELi4ELi64ELi3ELb0ELi0ELb1EEv10ScanParams')):
    a = next(i for i, l in enumerate(L) if l.startswith(kern + ':'))
    b = next(i for i in range(a, len(L)) if 's_endpgm' in L[i])
    K = L[a:b]
    open(os.path.join(ROOT, 'build/isa/%s.s' % tag), 'w').write('\n'.join(K))
    bars = [i for i, l in enumerate(K) if 's_barrier' in l]
    seg = max(zip(bars, bars[1:]), key=lambda ab: ab[1] - ab[0])
    r = [i for i in range(*seg) if 'v_frexp_mant_f64' in K[i]]
    gaps = sorted(y - x for x, y in zip(r, r[1:])); med = gaps[len(gaps) // 2]
    labels = [i for i in range(*seg) if re.match(r'\.LBB\d+_\d+:', K[i])]
    runs = []; cur = [r[0]]
    for x, y in zip(r, r[1:]):
        if y - x > 2.2 * med: runs.append(cur); cur = [y]
        else: cur.append(y)
    runs.append(cur)
    for run in runs:
        lo, hi = run[0], run[-1]
        n = max(1, len(run) - 1)
        c = lambda p: sum(1 for l in K[lo:hi] if re.match(r'\s+' + p, l))
        print(tag, 'copy at %5d..%5d: %2d positions; per position: valu %.1f salu %.1f lds %.1f vmem %.1f; scratch ops %d, labels %d' % (
            lo, hi, len(run), c('v_') / n, c('s_') / n, c('ds_') / n, c('(global|buffer|flat)_') / n, c('scratch_'), sum(1 for l in labels if lo < l < hi)))
