"""frisk_amd - MI355X (gfx950) implementation of the hot path of Adamtaranto/frisk.

Scope: genome k-mer profile (phase A) and the sliding-window scan
k-mer counts -> IVOM -> Kullback-Leibler score (+GC, RIP) (phase B), as hand-written HIP
kernels behind the C ABI of include/frisk_hip.h, called through ctypes.  There is no CPU
fallback: using the hot path without the built HIP library raises.
"""
from .engine import Engine, ScanResult, profile_len, table_offset  # noqa: F401

__version__ = "0.1.0"
