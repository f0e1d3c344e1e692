"""ScoreTable: the rows of the score table as COLUMNS (numpy arrays), the form the GPU hands them over in.

The reference appends one row at a time to a pandas DataFrame (`allWindows.loc[len(allWindows)] = row`, L1488: O(n) per
append, O(n^2) overall) and writes / prints each row as text (L1487-1494).  At GRCh38 scale that is 3 M rows: here the
columns stay numpy arrays end to end, the text comes from the library's native formatter (frisk_format_rows), and
only the few rows that post-processing selects (anomalies, RIP features) ever become Python tuples.
"""
import ctypes as C

import numpy as np

from . import _ffi

BASE_COLUMNS = ["name", "start", "stop", "windowKLD", "GC"]
RIP_COLUMNS = ["PI", "SI", "CRI"]


class ScoreTable:
    def __init__(self, names, seq_index, start, stop, kld, gc, pi=None, si=None, cri=None, kld_is_int0=None):
        self.names = list(names)                                    # one per scaffold
        self.seq_index = np.ascontiguousarray(seq_index, dtype=np.int32)
        self.start = np.ascontiguousarray(start, dtype=np.int64)
        self.stop = np.ascontiguousarray(stop, dtype=np.int64)
        self.kld = np.ascontiguousarray(kld, dtype=np.float64)
        self.gc = np.ascontiguousarray(gc, dtype=np.float64)
        self.rip = pi is not None
        self.pi = np.ascontiguousarray(pi, dtype=np.float64) if self.rip else None
        self.si = np.ascontiguousarray(si, dtype=np.float64) if self.rip else None
        self.cri = np.ascontiguousarray(cri, dtype=np.float64) if self.rip else None
        n = len(self.seq_index)
        self.kld_is_int0 = (np.zeros(n, np.uint8) if kld_is_int0 is None
                            else np.ascontiguousarray(kld_is_int0, dtype=np.uint8))   # empty sum: the int 0 (L465)

    # ------------------------------------------------------------------ construction
    @classmethod
    def from_rows(cls, rows, rip=None):
        """From a list of tuples (name, start, stop, KLD, GC[, PI, SI, CRI]) - caches, tests, small tables."""
        rows = list(rows)
        if rip is None:
            rip = bool(rows) and len(rows[0]) >= 8
        names, index = [], {}
        seq = np.zeros(len(rows), np.int32)
        for i, r in enumerate(rows):
            if r[0] not in index:
                index[r[0]] = len(names)
                names.append(r[0])
            seq[i] = index[r[0]]
        col = lambda j, dt: np.array([r[j] for r in rows], dtype=dt) if rows else np.zeros(0, dt)      # noqa: E731
        int0 = np.array([isinstance(r[3], int) and not isinstance(r[3], bool) and r[3] == 0 for r in rows], dtype=np.uint8)
        return cls(names, seq, col(1, np.int64), col(2, np.int64), col(3, np.float64), col(4, np.float64),
                   col(5, np.float64) if rip else None, col(6, np.float64) if rip else None,
                   col(7, np.float64) if rip else None, int0)

    @classmethod
    def from_frame(cls, frame, rip):
        """From the DataFrame the reference pickles (L1501), incl. the legacy windowKLI column name (L1458-1459)."""
        kcol = "windowKLD" if "windowKLD" in frame.columns else "windowKLI"
        cols = ["name", "start", "stop", kcol, "GC"] + (RIP_COLUMNS if rip else [])
        numeric = all(frame[c].dtype.kind in "iuf" for c in cols[1:])
        if not numeric:                 # object columns (a frame built row by row, as the reference's: the int 0 of L465 survives there)
            return cls.from_rows([tuple(r) for r in frame[cols].itertuples(index=False, name=None)], rip=rip)
        # the frame this package pickles: numeric columns - taken over as arrays (3 M rows: milliseconds instead of seconds)
        import pandas as pd
        codes, uniq = pd.factorize(frame["name"], sort=False)          # scaffolds in order of first appearance, as from_rows
        col = lambda c, dt: frame[c].to_numpy(dtype=dt)                 # noqa: E731
        return cls([str(u) for u in uniq], codes.astype(np.int32), col("start", np.int64), col("stop", np.int64), col(kcol, np.float64),
                   col("GC", np.float64), col("PI", np.float64) if rip else None, col("SI", np.float64) if rip else None,
                   col("CRI", np.float64) if rip else None, None)

    # ------------------------------------------------------------------ views
    def __len__(self):
        return int(self.seq_index.shape[0])

    @property
    def columns(self):
        return BASE_COLUMNS + (RIP_COLUMNS if self.rip else [])

    def name_column(self):
        return np.asarray(self.names, dtype=object)[self.seq_index] if len(self) else np.zeros(0, dtype=object)

    def row(self, r):
        kld = 0 if self.kld_is_int0[r] else float(self.kld[r])
        t = (self.names[int(self.seq_index[r])], int(self.start[r]), int(self.stop[r]), kld, float(self.gc[r]))
        if self.rip:
            t += (float(self.pi[r]), float(self.si[r]), float(self.cri[r]))
        return t

    def rows(self, index=None):
        """Python tuples - for the few rows that post-processing selects, for tests, for small tables."""
        idx = range(len(self)) if index is None else np.asarray(index).tolist()
        return [self.row(r) for r in idx]

    def head(self, n):
        sl = slice(0, n)
        return ScoreTable(self.names, self.seq_index[sl], self.start[sl], self.stop[sl], self.kld[sl], self.gc[sl],
                          self.pi[sl] if self.rip else None, self.si[sl] if self.rip else None,
                          self.cri[sl] if self.rip else None, self.kld_is_int0[sl])

    def to_frame(self):
        import pandas as pd
        data = {"name": self.name_column(), "start": self.start, "stop": self.stop,
                "windowKLD": np.where(self.kld_is_int0 != 0, 0.0, self.kld), "GC": self.gc}
        if self.rip:
            data.update(PI=self.pi, SI=self.si, CRI=self.cri)
        return pd.DataFrame(data, columns=self.columns, copy=False)      # (views of the table's columns: nothing is written to them)

    # ------------------------------------------------------------------ text
    def text(self, fmt=None):
        """The table body (no header): one line per row, tab separated, as the reference writes it (L1487-1494).
        fmt None = Python 2's str() through the native formatter; a callable = that per-value formatter (slow path)."""
        n = len(self)
        if n == 0:
            return ""
        if fmt is not None:
            return "".join("\t".join(fmt(v) for v in self.row(r)) + "\n" for r in range(n))
        return self.text_bytes().decode("utf-8", "replace")

    def text_bytes(self):
        """The table body as the native formatter's bytes (UTF-8; Python 2's str() for floats)."""
        n = len(self)
        if n == 0:
            return b""
        lib = _ffi.lib()
        names = (C.c_char_p * max(len(self.names), 1))(*[s.encode("utf-8", "replace") for s in self.names])
        p = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None      # noqa: E731
        out_len = C.c_int64()
        buf = lib.frisk_format_rows(n, names, p(self.seq_index), p(self.start), p(self.stop), p(self.kld_is_int0),
                                    p(self.kld), p(self.gc), p(self.pi), p(self.si), p(self.cri), C.byref(out_len))
        if not buf:
            raise MemoryError("frisk_format_rows failed")
        try:
            return C.string_at(buf, out_len.value)
        finally:
            lib.frisk_free(C.c_void_p(buf))
