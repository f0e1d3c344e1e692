#!/usr/bin/env python3
"""Experiment (GPU box): where frisk_fasta_load's time goes on an existing FASTA (parse / upload / pack)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from frisk_amd import Engine, _ffi
path = sys.argv[1]
t0 = time.perf_counter()
n, total, dig = C.c_int32(), C.c_int64(), C.c_uint64()
_ffi.lib().frisk_fasta_digest(os.fsencode(path), C.byref(n), C.byref(total), C.byref(dig))
t1 = time.perf_counter()
print("parse + digest (host only): %.3f s for %d bases" % (t1 - t0, total.value))
with Engine(1, 8) as e:
    for rep in range(2):
        t0 = time.perf_counter(); e.load_fasta(path); t1 = time.perf_counter()
        print("load_fasta #%d: %.3f s (pack kernel %.2f ms)" % (rep, t1 - t0, e.kernel_ms(2)))
