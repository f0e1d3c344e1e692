#!/usr/bin/env python3
"""(GPU box) where HotPath._load's time goes on the whole-C5 FASTA that tools/e2e_cli.py leaves in /tmp/frisk_e2e: the library's own
split (a -DFRISK_TUNE build prints parse / host pack / upload with FRISK_LOAD_SPLIT=1), then export_2bit (the host copy handed to the
sequence-cache writer) and Engine.close().  usage: FRISK_HIP_LIB=build/tune.so FRISK_LOAD_SPLIT=1 python tools/exp/load_split.py [fasta]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from frisk_amd.engine import Engine
fa = sys.argv[1] if len(sys.argv) > 1 else "/tmp/frisk_e2e/C5.fa"
for r in range(3):
    e = Engine(1, 8)
    t0 = time.perf_counter(); names = e.load_fasta(fa); t1 = time.perf_counter()
    out = e.export_2bit(); t2 = time.perf_counter()
    e.profile_reset(); e.profile_add(); e.profile_finalize(); res = e.scan(5000, 1000); t3 = time.perf_counter()
    e.close(); t4 = time.perf_counter()
    print("load_fasta %.1f ms, export_2bit %.1f ms, profile + scan %.1f ms, close %.1f ms (%d records)" % (
        (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, len(names)), flush=True)
