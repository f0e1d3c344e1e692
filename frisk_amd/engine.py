"""Engine: one libfrisk_hip context (one GPU, one HIP stream) with numpy in / numpy out.

Thin host plumbing over the C ABI (include/frisk_hip.h); all arithmetic of the hot path runs in
the HIP kernels.  torch is used only by `profile_allreduce` (RCCL through torch.distributed).
"""
import ctypes as C
import os

import numpy as np

from . import _ffi


def profile_len(kmin, kmax):
    return sum(4 ** x for x in range(kmin, kmax + 1))


def table_offset(kmin, x):
    return (4 ** x - 4 ** kmin) // 3


class ScanResult:
    """Per-candidate arrays of one frisk_scan call; `kept` selects the rows the reference emits."""

    def __init__(self, **kw):
        self.__dict__.update(kw)

    @property
    def kept(self):
        return (self.status & _ffi.ROW_KEPT) != 0

    @property
    def zero_weight(self):
        return (self.status & _ffi.ROW_ZERO_WEIGHT) != 0

    def __len__(self):
        return int(self.status.shape[0])


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def pack_2bit_host(seqs):
    """frisk_pack_2bit without a context (host-only entry point of the library): (codes, inv_runs, low_runs, lens) as ordinary
    numpy arrays - the 0.25 B/base form `Engine.stage_2bit` uploads."""
    lib = _ffi.lib()
    arrs = [Engine._as_u8(s) for s in seqs]
    n = len(arrs)
    ptrs = (C.c_void_p * max(n, 1))(*[a.ctypes.data for a in arrs])
    lens = [int(a.size) for a in arrs]
    clens = (C.c_int64 * max(n, 1))(*lens)
    P = int(lib.frisk_padded_len_of(clens, n))
    codes = np.empty(2 * P // 32, np.uint32)
    pi, pl, ni, nl = C.c_void_p(), C.c_void_p(), C.c_int64(), C.c_int64()
    rc = lib.frisk_pack_2bit(ptrs, clens, n, _ptr(codes), C.byref(pi), C.byref(ni), C.byref(pl), C.byref(nl))
    if rc != _ffi.OK:
        raise _ffi.FriskHipError(rc, "frisk_pack_2bit failed")
    runs = []
    for ptr, cnt in ((pi, ni), (pl, nl)):
        k = int(cnt.value)
        a = np.frombuffer((C.c_int64 * (2 * k)).from_address(ptr.value), dtype=np.int64).copy() if k else np.empty(0, np.int64)
        lib.frisk_free(ptr)
        runs.append(a.reshape(k, 2))
    return codes, runs[0], runs[1], lens


def fasta_pack_2bit_host(path):
    """frisk_fasta_pack_2bit (host-only): a FASTA file straight into the 0.25 B/base form: (codes, inv_runs, low_runs, lens)."""
    lib = _ffi.lib()
    n = C.c_int32()
    pl, pc, pi, pw = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
    nc, ni, nw = C.c_int64(), C.c_int64(), C.c_int64()
    rc = lib.frisk_fasta_pack_2bit(os.fsencode(path), C.byref(n), C.byref(pl), C.byref(pc), C.byref(nc), C.byref(pi), C.byref(ni),
                                   C.byref(pw), C.byref(nw))
    if rc != _ffi.OK:
        raise _ffi.FriskHipError(rc, "frisk_fasta_pack_2bit failed: %s" % path)

    def take(ptr, count, ctype, dtype):
        a = np.frombuffer((ctype * count).from_address(ptr.value), dtype=dtype).copy() if count else np.empty(0, dtype)
        lib.frisk_free(ptr)
        return a
    lens = [int(x) for x in take(pl, int(n.value), C.c_int64, np.int64)]
    codes = take(pc, int(nc.value), C.c_uint32, np.uint32)
    inv = take(pi, 2 * int(ni.value), C.c_int64, np.int64).reshape(-1, 2)
    low = take(pw, 2 * int(nw.value), C.c_int64, np.int64).reshape(-1, 2)
    return codes, inv, low, lens


class Engine:
    def __init__(self, kmin, kmax, device=0):
        self._lib = _ffi.lib()
        self._ctx = C.c_void_p()
        self.kmin, self.kmax, self.device = int(kmin), int(kmax), int(device)
        rc = self._lib.frisk_create(self.device, self.kmin, self.kmax, C.byref(self._ctx))
        if rc != _ffi.OK:
            msg = self._lib.frisk_last_error(self._ctx).decode() if self._ctx else "allocation failed"
            if self._ctx:
                self._lib.frisk_destroy(self._ctx)
                self._ctx = C.c_void_p()
            raise _ffi.FriskHipError(rc, msg)
        self.nprof = int(self._lib.frisk_profile_len(self._ctx))
        self.n_seq = 0
        self.seq_lens = []
        self.shard_index = None     # the seek index the last load_fasta_shard used (None: it parsed the file)
        self._pinned = {}           # name -> (address, nbytes): page-locked result buffers, reused across scans

    # ------------------------------------------------------------------ lifetime
    def close(self):
        if getattr(self, "_ctx", None):
            for addr, _ in self._pinned.values():
                self._lib.frisk_host_free(self._ctx, C.c_void_p(addr))
            self._pinned = {}
            self._lib.frisk_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def _pinned_array(self, name, n, dtype):
        """numpy view of a page-locked buffer owned by this engine (grown on demand, reused between calls)."""
        dtype = np.dtype(dtype)
        need = max(int(n), 1) * dtype.itemsize
        addr, have = self._pinned.get(name, (0, 0))
        if have < need:
            if addr:
                self._lib.frisk_host_free(self._ctx, C.c_void_p(addr))
            cap = need + need // 8
            addr = self._lib.frisk_host_alloc(self._ctx, cap)
            if not addr:
                raise MemoryError("frisk_host_alloc(%d) failed" % cap)
            self._pinned[name] = (addr, cap)
        buf = (C.c_char * need).from_address(addr)
        return np.frombuffer(buf, dtype=dtype, count=max(int(n), 1))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc):
        if rc != _ffi.OK:
            raise _ffi.FriskHipError(rc, self._lib.frisk_last_error(self._ctx).decode())

    # ----------------------------------------------------------------- sequences
    def load(self, seqs):
        """Make a batch of scaffolds resident (list of bytes / str / uint8 arrays)."""
        bufs = []
        for s in seqs:
            if isinstance(s, str):
                s = s.encode("ascii")
            elif isinstance(s, np.ndarray):
                s = s.astype(np.uint8, copy=False).tobytes()
            bufs.append(bytes(s))
        n = len(bufs)
        arr = (C.c_char_p * max(n, 1))(*bufs)
        lens = (C.c_int64 * max(n, 1))(*[len(b) for b in bufs])
        self._check(self._lib.frisk_seq_load(self._ctx, arr, lens, n))
        self.n_seq = n
        self.seq_lens = [len(b) for b in bufs]

    def host_array(self, name, n, dtype=np.uint8):
        """numpy array in page-locked host memory owned by this engine (uploads from it are asynchronous)."""
        return self._pinned_array("host:" + name, n, dtype)

    @staticmethod
    def _as_u8(s):
        if isinstance(s, np.ndarray):
            return np.ascontiguousarray(s, dtype=np.uint8)
        if isinstance(s, str):
            s = s.encode("ascii")
        return np.frombuffer(bytes(s), dtype=np.uint8)

    def stage(self, seqs):
        """Start uploading + packing the NEXT batch on the copy stream while the resident one is in use; `commit()` makes it
        resident.  Asynchronous when the scaffolds are views of `host_array` buffers; keep them alive until after commit."""
        arrs = [self._as_u8(s) for s in seqs]
        n = len(arrs)
        ptrs = (C.c_void_p * max(n, 1))(*[a.ctypes.data for a in arrs])
        lens = (C.c_int64 * max(n, 1))(*[a.size for a in arrs])
        self._check(self._lib.frisk_seq_stage(self._ctx, ptrs, lens, n))
        self._staged = (arrs, [int(a.size) for a in arrs])

    def stage_packed(self, codes, inv, low, lens):
        """The same from the three bit-packed arrays (library layout, as `export_packed` returns them): 0.5 B/base."""
        lens = [int(x) for x in lens]
        arr = (C.c_int64 * max(len(lens), 1))(*lens)
        keep = [np.ascontiguousarray(a, dtype=np.uint32) for a in (codes, inv, low)]
        self._check(self._lib.frisk_seq_stage_packed(self._ctx, _ptr(keep[0]), _ptr(keep[1]), _ptr(keep[2]), arr, len(lens)))
        self._staged = (keep, lens)

    def pack_2bit(self, seqs, pinned=True):
        """ASCII scaffolds -> the 0.25 B/base upload form, on the host (library threads): (codes, inv_runs, low_runs, lens).
        codes: uint32[2 P / 32] (page-locked when pinned=True: its upload is then asynchronous); inv_runs / low_runs: int64[n, 2]
        half-open runs of padded positions (letters other than ACGTacgt / lowercase acgt)."""
        arrs = [self._as_u8(s) for s in seqs]
        n = len(arrs)
        ptrs = (C.c_void_p * max(n, 1))(*[a.ctypes.data for a in arrs])
        lens = [int(a.size) for a in arrs]
        clens = (C.c_int64 * max(n, 1))(*lens)
        P = int(self._lib.frisk_padded_len_of(clens, n))
        codes = self.host_array("codes2", 2 * P // 32, np.uint32) if pinned else np.empty(2 * P // 32, np.uint32)
        pi, pl, ni, nl = C.c_void_p(), C.c_void_p(), C.c_int64(), C.c_int64()
        self._check(self._lib.frisk_pack_2bit(ptrs, clens, n, _ptr(codes), C.byref(pi), C.byref(ni), C.byref(pl), C.byref(nl)))
        return (codes,) + self._take_runs(pi, ni, pl, nl, pinned) + (lens,)

    def _take_runs(self, pi, ni, pl, nl, pinned):
        out = []
        for tag, ptr, cnt in (("inv_runs", pi, ni), ("low_runs", pl, nl)):
            k = int(cnt.value)
            dst = self.host_array(tag, max(2 * k, 2), np.int64)[:2 * k] if pinned else np.empty(2 * k, np.int64)
            if k:
                dst[:] = np.frombuffer((C.c_int64 * (2 * k)).from_address(ptr.value), dtype=np.int64)
            self._lib.frisk_free(ptr)
            out.append(dst.reshape(k, 2))
        return tuple(out)

    def stage_2bit(self, codes, inv_runs, low_runs, lens, piece_bases=0):
        """Start the streamed upload of the NEXT batch from the 0.25 B/base form (`pack_2bit` / `export_2bit`); `commit()` makes
        it resident without waiting, and `profile_add()` on it follows the pieces of the upload.  inv_runs / low_runs: int64[n, 2]
        run lists, or a uint32[P / 32] dense bitmap."""
        lens = [int(x) for x in lens]
        arr = (C.c_int64 * max(len(lens), 1))(*lens)
        keep = [np.ascontiguousarray(codes, dtype=np.uint32)]
        args = []
        for m in (inv_runs, low_runs):
            m = np.asarray(m)
            if m.dtype == np.uint32 and m.ndim == 1:            # dense bitmap
                m = np.ascontiguousarray(m)
                args += [_ptr(m), -1]
            else:
                m = np.ascontiguousarray(m, dtype=np.int64).reshape(-1, 2)
                args += [_ptr(m), int(m.shape[0])]
            keep.append(m)
        self._check(self._lib.frisk_seq_stage_2bit(self._ctx, _ptr(keep[0]), args[0], args[1], args[2], args[3], arr, len(lens),
                                                   int(piece_bases)))
        self._staged = (keep, lens)

    def export_2bit(self, pinned=False):
        """(codes, inv_runs, low_runs) of the resident batch in the 0.25 B/base form."""
        w32 = self.padded_len // 32
        codes = self.host_array("codes2", 2 * w32, np.uint32) if pinned else np.empty(2 * w32, np.uint32)
        pi, pl, ni, nl = C.c_void_p(), C.c_void_p(), C.c_int64(), C.c_int64()
        self._check(self._lib.frisk_seq_export_2bit(self._ctx, _ptr(codes), C.byref(pi), C.byref(ni), C.byref(pl), C.byref(nl)))
        return (codes,) + self._take_runs(pi, ni, pl, nl, pinned)

    def commit(self, names=None):
        self._check(self._lib.frisk_seq_commit(self._ctx))
        self._keepalive, lens = self._staged
        self.n_seq, self.seq_lens = len(lens), list(lens)
        if names is not None:
            arr = (C.c_char_p * max(len(names), 1))(*[n.encode("ascii", "replace") for n in names])
            self._check(self._lib.frisk_seq_set_names(self._ctx, arr, len(names)))

    def export_packed(self, pinned=False):
        """(codes, inv, low) of the resident batch as uint32 arrays (2P/32, P/32, P/32 words; P = padded_len)."""
        w32 = self.padded_len // 32
        new = (lambda nm, n: self.host_array(nm, n, np.uint32)) if pinned else (lambda nm, n: np.empty(n, np.uint32))
        codes, inv, low = new("codes", 2 * w32), new("inv", w32), new("low", w32)
        self._check(self._lib.frisk_seq_export_packed(self._ctx, _ptr(codes), _ptr(inv), _ptr(low)))
        return codes, inv, low

    def load_fasta(self, path):
        """Parse a FASTA / FASTA.gz file in the library (no Python per-line loop) and make its records resident.
        Returns the record names."""
        n, total = C.c_int32(), C.c_int64()
        self._check(self._lib.frisk_fasta_load(self._ctx, os.fsencode(path), C.byref(n), C.byref(total)))
        self.n_seq = int(n.value)
        self.seq_lens = [int(self._lib.frisk_seq_len(self._ctx, i)) for i in range(self.n_seq)]
        return [self._lib.frisk_seq_name(self._ctx, i).decode("ascii", "replace") for i in range(self.n_seq)]

    def load_fasta_shard(self, path, w, inc, rank, world, scaffolds_all=False, index=None):
        """One rank's share of a multi-GPU job (window tiles + halo): keep resident only the bases of the rank's candidate
        windows and of the positions it counts.  With a usable seek index (`index`: a path or a list of paths to try,
        fasta.fastaIndexPaths; written by fasta.writeFastaIndex) the rank copies just those bytes from the mapped file;
        without one it parses the whole file natively.  `self.shard_index` = the index that was used, or None.
        Returns (names of ALL records, (cand_begin, cand_end)): scan() on this batch numbers candidates from 0 = cand_begin."""
        n, total, c0, c1 = C.c_int32(), C.c_int64(), C.c_int64(), C.c_int64()
        flags = _ffi.SCAN_SCAFFOLDS_ALL if scaffolds_all else 0
        self.shard_index = None
        for cand in ([index] if isinstance(index, (str, bytes, os.PathLike)) else list(index or [])):
            if not os.path.exists(cand):
                continue
            rc = self._lib.frisk_fasta_load_shard_indexed(self._ctx, os.fsencode(path), os.fsencode(cand), int(w), int(inc), flags,
                                                          int(rank), int(world), C.byref(n), C.byref(total), C.byref(c0), C.byref(c1))
            if rc == _ffi.E_INDEX:
                continue                # (not this file's index, or a file the byte arithmetic cannot address: parse instead)
            self._check(rc)
            self.shard_index = os.fspath(cand)
            break
        if self.shard_index is None:
            self._check(self._lib.frisk_fasta_load_shard(self._ctx, os.fsencode(path), int(w), int(inc), flags, int(rank), int(world),
                                                         C.byref(n), C.byref(total), C.byref(c0), C.byref(c1)))
        self.n_seq = int(n.value)
        self.seq_lens = [int(self._lib.frisk_seq_len(self._ctx, i)) for i in range(self.n_seq)]
        names = [self._lib.frisk_seq_name(self._ctx, i).decode("ascii", "replace") for i in range(self.n_seq)]
        return names, (int(c0.value), int(c1.value))

    def synth(self, lens, seed, island_frac=0.02, n_frac=0.0, lower_frac=0.0, repeats_per_kb=0.0, period_mix=0.0, sat_frac=0.0):
        lens = [int(x) for x in lens]
        arr = (C.c_int64 * max(len(lens), 1))(*lens)
        self._check(self._lib.frisk_seq_synth2(self._ctx, arr, len(lens), C.c_uint64(seed), float(island_frac),
                                               float(n_frac), float(lower_frac), float(repeats_per_kb), float(period_mix),
                                               float(sat_frac)))
        self.n_seq = len(lens)
        self.seq_lens = lens

    def read_seq(self, index, offset=0, n=None):
        if n is None:
            n = self.seq_lens[index] - offset
        out = np.empty(max(n, 1), dtype=np.uint8)
        self._check(self._lib.frisk_seq_read(self._ctx, index, int(offset), int(n), _ptr(out)))
        return out[:n].tobytes()

    @property
    def padded_len(self):
        return int(self._lib.frisk_seq_padded_len(self._ctx))

    # ------------------------------------------------------------------- phase A
    def profile_reset(self):
        self._check(self._lib.frisk_profile_reset(self._ctx))

    def profile_add(self, mask_host=False, pos_begin=-1, pos_end=-1, one_pass=False):
        """one_pass=True (test hook, kmax = 8): the 16-bit one-pass form that long ranges take by themselves, on any size."""
        self._check(self._lib.frisk_profile_add(self._ctx, (1 if mask_host else 0) | (2 if one_pass else 0), pos_begin, pos_end))

    def profile_raw(self):
        out = np.empty(self.nprof + 4, dtype=np.int64)
        self._check(self._lib.frisk_profile_export_host(self._ctx, _ptr(out)))
        return out

    def profile_set_raw(self, raw):
        raw = np.ascontiguousarray(raw, dtype=np.int64)
        assert raw.shape == (self.nprof + 4,)
        self._check(self._lib.frisk_profile_import_host(self._ctx, _ptr(raw)))

    def profile_allreduce(self, group=None, force=False, comm=None):
        """Sum the raw (linear) profile over all ranks: the ONE collective of a job.  `self.allreduce_path` names what ran:
          rccl_direct               comm = an ncclComm_t (int / c_void_p): the library's own frisk_profile_allreduce, no torch
          in_place_external_stream  torch.distributed with the nccl (= RCCL) backend, IN PLACE on the library's buffer under the
                                    context's stream - no copies, no host wait.  The first use in a group of several ranks is
                                    CHECKED against the copying path below (one extra all-reduce, once per engine); if
                                    ProcessGroupNCCL rejects the zero-copy view or the external stream, or the results
                                    differ, the engine falls back for good
          export_import_copy        export to a torch tensor -> all_reduce -> import (two D2D copies, host waits)
          host_gloo                 any backend that reduces CPU tensors (the CPU rehearsal of the multi-rank path)
        force=True runs the collective even in a one-rank group (tests)."""
        if comm is not None:
            self._check(self._lib.frisk_profile_allreduce(self._ctx, C.c_void_p(int(getattr(comm, "value", comm) or 0))))
            self.allreduce_path = "rccl_direct"
            return
        import torch
        import torch.distributed as dist
        if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force):
            return
        n = self.nprof + 4
        if dist.get_backend(group) != "nccl":
            from .distributed import allreduce_raw_host
            self.profile_set_raw(allreduce_raw_host(self.profile_raw(), group))
            self.allreduce_path = "host_gloo"
            return
        dev = torch.device("cuda", self.device)

        def copying():
            t = torch.empty(n, dtype=torch.int64, device=dev)
            self._check(self._lib.frisk_profile_export_device(self._ctx, C.c_void_p(t.data_ptr())))
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            torch.cuda.synchronize(dev)
            self._check(self._lib.frisk_profile_import_device(self._ctx, C.c_void_p(t.data_ptr())))
            return t

        def in_place():
            # the library's raw-profile buffer as a torch tensor, the collective issued under the context's stream
            # (ProcessGroupNCCL chains its own stream to the current one with events, both ways): profile_add -> all-reduce ->
            # finalize stay one chain of enqueued work.  The view and the stream wrapper are kept from step to step.
            raw, stream = C.c_void_p(), C.c_void_p()
            self._check(self._lib.frisk_profile_device_view(self._ctx, C.byref(raw), C.byref(stream)))
            key = (int(raw.value), int(stream.value or 0), n)
            kept = getattr(self, "_allreduce_keep", None)
            if kept is None or kept[0] != key:
                class _View:
                    __cuda_array_interface__ = {"shape": (n,), "typestr": "<i8", "data": (key[0], False), "version": 2}
                kept = (key, torch.as_tensor(_View(), device=dev), torch.cuda.ExternalStream(key[1], device=dev))
                self._allreduce_keep = kept
            _, t, ext = kept
            with torch.cuda.stream(ext):
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            return t

        mode = getattr(self, "_allreduce_mode", None)
        if mode == "copy":
            copying()
            self.allreduce_path = "export_import_copy"
            return
        if mode == "in_place" or dist.get_world_size(group) == 1:
            in_place()
            self.allreduce_path = "in_place_external_stream"
            return
        # first use with several ranks: both paths on the same input, compared on the device; every rank takes the same decision
        import logging
        before = torch.empty(n, dtype=torch.int64, device=dev)
        self._check(self._lib.frisk_profile_export_device(self._ctx, C.c_void_p(before.data_ptr())))
        ok = 1
        try:
            fast = in_place().clone()
            torch.cuda.synchronize(dev)
        except Exception as err:                 # noqa: BLE001 - whatever ProcessGroupNCCL objects to, the copying path does not need
            logging.getLogger("frisk_amd").warning("in-place all-reduce refused (%s): copying path from now on", err)
            ok, fast = 0, None
        self._check(self._lib.frisk_profile_import_device(self._ctx, C.c_void_p(before.data_ptr())))
        slow = copying()
        if ok and not bool(torch.equal(fast, slow)):
            logging.getLogger("frisk_amd").warning("in-place all-reduce disagrees with the copying path: copying path from now on")
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int64, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        self._allreduce_mode = "in_place" if int(flag.item()) == 1 else "copy"
        self.allreduce_path = "export_import_copy (first use: in-place path %s)" % ("verified" if self._allreduce_mode == "in_place" else "rejected")

    def profile_finalize(self):
        self._check(self._lib.frisk_profile_finalize(self._ctx))

    def profile_get(self):
        sym = np.empty(self.nprof, dtype=np.int64)
        tl, ex, nn = C.c_int64(), C.c_int64(), C.c_int64()
        self._check(self._lib.frisk_profile_get(self._ctx, _ptr(sym), C.byref(tl), C.byref(ex), C.byref(nn)))
        return sym, int(tl.value), int(ex.value), int(nn.value)

    def profile_set(self, sym, total_len, ex_max, nn_total):
        sym = np.ascontiguousarray(sym, dtype=np.int64)
        assert sym.shape == (self.nprof,)
        self._check(self._lib.frisk_profile_set(self._ctx, _ptr(sym), int(total_len), int(ex_max), int(nn_total)))

    # ------------------------------------------------------------------- phase B
    def scan_plan(self, w, inc, scaffolds_all=False):
        n = C.c_int64()
        flags = _ffi.SCAN_SCAFFOLDS_ALL if scaffolds_all else 0
        self._check(self._lib.frisk_scan_plan(self._ctx, int(w), int(inc), flags, C.byref(n)))
        return int(n.value)

    def scan(self, w, inc, rip=False, scaffolds_all=False, c0=0, c1=-1, debug=False, pinned=False, chunks=False, bits4=False, side4=False):
        """Score candidates [c0, c1).  chunks=True: the schedule of a long scan (chunks of 8 windows, tables sliding inside a
        chunk) whatever the size; bits4=True: K = 8 starts with 4-bit counters whatever the size, side4=True: ... with the side table for max-mers of period <= 4 - same rows either way.  With pinned=True the result arrays are views of page-locked buffers owned
        by the engine: D2H at PCIe rate and no per-call allocation, but the views are only valid until the next
        scan() on this engine.  pinned=False (default) returns ordinary numpy arrays."""
        flags = (_ffi.SCAN_RIP if rip else 0) | (_ffi.SCAN_SCAFFOLDS_ALL if scaffolds_all else 0) | (_ffi.SCAN_CHUNKS if chunks else 0) | \
                (_ffi.SCAN_BITS4 if bits4 else 0) | (_ffi.SCAN_SIDE4 if side4 else 0)
        total = self.scan_plan(w, inc, scaffolds_all)
        if c1 < 0:
            c1 = total
        n = max(c1 - c0, 0)
        cap = max(n, 1)
        if pinned and not debug:
            new = lambda name, dt: self._pinned_array(name, cap, dt)       # noqa: E731
        else:
            new = lambda name, dt: np.zeros(cap, dt)                       # noqa: E731
        r = ScanResult(
            seq_index=new("seq", np.int32), start=new("start", np.int64), stop=new("stop", np.int64),
            status=new("status", np.uint32), kld=new("kld", np.float64), gc=new("gc", np.float64),
            pi=new("pi", np.float64) if rip else None, si=new("si", np.float64) if rip else None,
            cri=new("cri", np.float64) if rip else None,
            counts=np.zeros((cap, self.nprof), np.uint32) if debug else None,
            meta=np.zeros((cap, 3), np.int64) if debug else None)
        self._check(self._lib.frisk_scan(self._ctx, int(w), int(inc), flags, int(c0), int(c1), cap,
                                         _ptr(r.seq_index), _ptr(r.start), _ptr(r.stop), _ptr(r.status),
                                         _ptr(r.kld), _ptr(r.gc), _ptr(r.pi), _ptr(r.si), _ptr(r.cri),
                                         _ptr(r.counts), _ptr(r.meta)))
        for k, v in list(r.__dict__.items()):
            if isinstance(v, np.ndarray):
                setattr(r, k, v[:n])
        r.n_candidates = total
        return r

    def scan_ivom(self, w, inc, scaffolds_all=False, c0=0, c1=-1):
        """(window_ivom, genome_ivom): IvomBuild's two normalised distributions per candidate window as dense
        [n, 4^kmax] arrays (test utility, kmax <= 6)."""
        flags = _ffi.SCAN_SCAFFOLDS_ALL if scaffolds_all else 0
        total = self.scan_plan(w, inc, scaffolds_all)
        if c1 < 0:
            c1 = total
        n = max(c1 - c0, 0)
        wi = np.zeros((max(n, 1), 4 ** self.kmax), np.float64)
        gi = np.zeros_like(wi)
        self._check(self._lib.frisk_scan_ivom(self._ctx, int(w), int(inc), flags, int(c0), int(c1), max(n, 1), _ptr(wi), _ptr(gi)))
        return wi[:n], gi[:n]

    def scan_stat(self):
        """(counter width of the bulk launch, windows handed 4->8 bit, windows handed on to 16 bit, row segments) of the last scan."""
        return tuple(int(self._lib.frisk_last_scan_stat(self._ctx, i)) for i in range(4))

    def scan_side(self):
        """True when the last scan's 4-bit bulk launch counted the max-mers of period <= 4 in its side table."""
        return int(self._lib.frisk_last_scan_stat(self._ctx, 4)) == 1

    def kernel_ms(self, which=0):
        return float(self._lib.frisk_last_kernel_ms(self._ctx, which))
