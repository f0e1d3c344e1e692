#!/bin/bash
# usage: run_nt.sh <lib.so> "<nt list>" "<bits list>" : NT x bits grid (C5/8 shard line only)
lib=$1
for bits in $3; do for nt in $2; do
  echo "== $lib NT=$nt bits=$bits"
  FRISK_K8_NT=$nt FRISK_HIP_LIB=$PWD/$lib timeout -k 10 200 python3 tools/exp/k8_variants.py 1.0 $bits 2>&1 | grep -v amdgpu.ids | python3 -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: print(l.rstrip()); continue
    if d['config'].startswith('C5'): print('%-28s bits %s  %.3f ms  %.2f Mwin/s  stat %s  sum %.12f'%(d['config'],d['bits'],d['scan_ms'],d['Mwin_per_s'],d['overflow'],d['kld_sum']))"
done; done
