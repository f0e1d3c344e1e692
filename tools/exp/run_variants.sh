#!/bin/bash
# usage: run_variants.sh <scale> <bits-list> lib1.so lib2.so ...   (GPU box) - k8_variants.py for each experiment library
scale=$1; bits=$2; shift 2
for lib in "$@"; do
  echo "== $lib"
  FRISK_HIP_LIB=$PWD/$lib timeout -k 10 200 python3 tools/exp/k8_variants.py $scale $bits 2>&1 | grep -v amdgpu.ids | python3 -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: print(l.rstrip()); continue
    print('%-28s bits %s  %.3f ms  %.2f Mwin/s  stat %s  diff %.2e  sum %.12f'%(d['config'],d['bits'],d['scan_ms'],d['Mwin_per_s'],d['overflow'],d['max_abs_diff_vs_first'] or 0,d['kld_sum']))"
done
