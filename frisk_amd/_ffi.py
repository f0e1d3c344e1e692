"""ctypes binding of libfrisk_hip.so (C ABI: include/frisk_hip.h).

The product path has no CPU fallback: if the HIP library is missing or cannot be loaded this
module raises at first use, loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FRISK_HIP_LIB") or os.path.join(_HERE, "libfrisk_hip.so")   # override: kernel experiments

OK, E_ARG, E_HIP, E_STATE, E_CAP, E_ZERO_WEIGHT, E_INDEX = 0, -1, -2, -3, -4, -5, -6
SCAN_RIP, SCAN_SCAFFOLDS_ALL, SCAN_CHUNKS, SCAN_BITS4, SCAN_SIDE4 = 1, 2, 256, 512, 1024
ROW_KEPT, ROW_ZERO_WEIGHT, ROW_JUMPBACK, ROW_NO_MAXMER = 1, 2, 4, 8

# every symbol include/frisk_hip.h declares: (name, restype, argtypes)
_P = C.c_void_p
_I64P = C.POINTER(C.c_int64)
SYMBOLS = [
    ("frisk_version", C.c_char_p, []),
    ("frisk_supported", C.c_int, [C.c_int, C.c_int, C.c_int64]),
    ("frisk_create", C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    ("frisk_destroy", None, [_P]),
    ("frisk_last_error", C.c_char_p, [_P]),
    ("frisk_profile_len", C.c_int64, [_P]),
    ("frisk_seq_load", C.c_int, [_P, C.POINTER(C.c_char_p), _I64P, C.c_int32]),
    ("frisk_seq_stage", C.c_int, [_P, C.POINTER(C.c_void_p), _I64P, C.c_int32]),
    ("frisk_seq_stage_packed", C.c_int, [_P, _P, _P, _P, _I64P, C.c_int32]),
    ("frisk_seq_stage_2bit", C.c_int, [_P, _P, _P, C.c_int64, _P, C.c_int64, _I64P, C.c_int32, C.c_int64]),
    ("frisk_pack_2bit", C.c_int, [C.POINTER(C.c_void_p), _I64P, C.c_int32, _P, C.POINTER(C.c_void_p), _I64P, C.POINTER(C.c_void_p), _I64P]),
    ("frisk_padded_len_of", C.c_int64, [_I64P, C.c_int32]),
    ("frisk_seq_export_2bit", C.c_int, [_P, _P, C.POINTER(C.c_void_p), _I64P, C.POINTER(C.c_void_p), _I64P]),
    ("frisk_seq_commit", C.c_int, [_P]),
    ("frisk_seq_export_packed", C.c_int, [_P, _P, _P, _P]),
    ("frisk_seq_set_names", C.c_int, [_P, C.POINTER(C.c_char_p), C.c_int32]),
    ("frisk_fasta_load", C.c_int, [_P, C.c_char_p, C.POINTER(C.c_int32), _I64P]),
    ("frisk_fasta_digest", C.c_int, [C.c_char_p, C.POINTER(C.c_int32), _I64P, C.POINTER(C.c_uint64)]),
    ("frisk_fasta_pack_2bit", C.c_int, [C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _I64P,
                               C.POINTER(C.c_void_p), _I64P, C.POINTER(C.c_void_p), _I64P]),
    ("frisk_fasta_load_shard", C.c_int, [_P, C.c_char_p, C.c_int32, C.c_int32, C.c_uint32, C.c_int32, C.c_int32,
                                         C.POINTER(C.c_int32), _I64P, _I64P, _I64P]),
    ("frisk_fasta_load_shard_indexed", C.c_int, [_P, C.c_char_p, C.c_char_p, C.c_int32, C.c_int32, C.c_uint32, C.c_int32, C.c_int32,
                                                 C.POINTER(C.c_int32), _I64P, _I64P, _I64P]),
    ("frisk_fasta_index_build", C.c_int, [C.c_char_p, C.c_char_p, C.POINTER(C.c_int32), C.c_char_p, C.c_int32]),
    ("frisk_fasta_index_read", C.c_int, [C.c_char_p, C.c_char_p, C.c_int32, C.c_int64, C.c_int64, _P, C.POINTER(C.c_int32), _I64P,
                                         C.c_char_p, C.c_int32, C.c_char_p, C.c_int32]),
    ("frisk_seq_count", C.c_int32, [_P]),
    ("frisk_seq_name", C.c_char_p, [_P, C.c_int32]),
    ("frisk_seq_len", C.c_int64, [_P, C.c_int32]),
    ("frisk_seq_synth", C.c_int, [_P, _I64P, C.c_int32, C.c_uint64, C.c_double, C.c_double, C.c_double, C.c_double]),
    ("frisk_seq_synth2", C.c_int, [_P, _I64P, C.c_int32, C.c_uint64, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                                   C.c_double]),
    ("frisk_seq_read", C.c_int, [_P, C.c_int32, C.c_int64, C.c_int64, _P]),
    ("frisk_profile_reset", C.c_int, [_P]),
    ("frisk_profile_add", C.c_int, [_P, C.c_int, C.c_int64, C.c_int64]),
    ("frisk_seq_padded_len", C.c_int64, [_P]),
    ("frisk_profile_raw_len", C.c_int64, [_P]),
    ("frisk_profile_export_device", C.c_int, [_P, _P]),
    ("frisk_profile_import_device", C.c_int, [_P, _P]),
    ("frisk_profile_device_view", C.c_int, [_P, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    ("frisk_profile_allreduce", C.c_int, [_P, _P]),
    ("frisk_profile_export_host", C.c_int, [_P, _P]),
    ("frisk_profile_import_host", C.c_int, [_P, _P]),
    ("frisk_profile_finalize", C.c_int, [_P]),
    ("frisk_profile_get", C.c_int, [_P, _P, _I64P, _I64P, _I64P]),
    ("frisk_profile_set", C.c_int, [_P, _P, C.c_int64, C.c_int64, C.c_int64]),
    ("frisk_scan_plan", C.c_int, [_P, C.c_int32, C.c_int32, C.c_uint32, _I64P]),
    ("frisk_scan", C.c_int, [_P, C.c_int32, C.c_int32, C.c_uint32, C.c_int64, C.c_int64, C.c_int64,
                             _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    ("frisk_scan_ivom", C.c_int, [_P, C.c_int32, C.c_int32, C.c_uint32, C.c_int64, C.c_int64, C.c_int64, _P, _P]),
    ("frisk_last_scan_stat", C.c_int64, [_P, C.c_int]),
    ("frisk_format_rows", C.c_void_p, [C.c_int64, C.POINTER(C.c_char_p), _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64P]),
    ("frisk_free", None, [_P]),
    ("frisk_hmm_fit", C.c_int, [_P, C.c_int64, C.c_int32, C.c_double, C.c_double, C.c_double, _P, _P, _P, _P,
                                C.POINTER(C.c_double), C.POINTER(C.c_int32)]),
    ("frisk_hmm_viterbi", C.c_int, [_P, _P, C.c_int32, _P, _P, _P, _P, _P]),
    ("frisk_host_alloc", C.c_void_p, [_P, C.c_int64]),
    ("frisk_host_free", None, [_P, _P]),
    ("frisk_last_kernel_ms", C.c_double, [_P, C.c_int]),
]

_lib = None


class FriskHipError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("libfrisk_hip error %d: %s" % (code, message))
        self.code = code


def lib():
    """Load the library once; raise (never fall back) if it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s not found: the HIP extension has not been built "
                "(run `python -c 'import __graft_entry__ as g; g.build()'`). "
                "frisk_amd has no CPU fallback." % LIB_PATH)
        # PyTorch-ROCm bundles its own HIP/HSA runtime; libfrisk_hip.so links the system one under the same
        # SONAME.  Whichever is loaded first serves both, and torch cannot see the GPU behind the system
        # runtime ("No HIP GPUs are available") - so when torch is installed, let it load its runtime first.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        handle = C.CDLL(LIB_PATH)
        for name, restype, argtypes in SYMBOLS:
            fn = getattr(handle, name)          # AttributeError if the .so lacks a declared symbol
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = handle
    return _lib
