"""ctypes binding of the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may import
this package; nothing under spumoni_amd/ does.  See oracle/spumoni_oracle.h.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("spumoni_oracle.c", "spumoni_oracle_t1.c", "orc_digest.c", "orc_queries.inc",
                                              "spumoni_oracle.h")]
    stale = (not os.path.exists(so)) or any(os.path.getmtime(f) > os.path.getmtime(so) for f in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        vp, u64, i32 = C.c_void_p, C.c_uint64, C.c_int
        L.orc_build.restype = vp
        L.orc_build.argtypes = [vp, vp, vp, u64, vp, vp, vp, vp]
        L.orc_load_raw.restype = vp
        L.orc_load_raw.argtypes = [C.c_char_p, i32]
        L.orc_free.argtypes = [vp]
        for name in ("orc_run_of_position",):
            getattr(L, name).restype = u64
            getattr(L, name).argtypes = [vp, u64]
        L.orc_at.restype = C.c_uint8
        L.orc_at.argtypes = [vp, u64]
        L.orc_rank.restype = u64
        L.orc_rank.argtypes = [vp, u64, C.c_uint8]
        L.orc_select.restype = u64
        L.orc_select.argtypes = [vp, u64, C.c_uint8]
        L.orc_run_head_rank.restype = u64
        L.orc_run_head_rank.argtypes = [vp, u64, C.c_uint8]
        L.orc_threshold.restype = u64
        L.orc_threshold.argtypes = [vp, u64]
        L.orc_LF.restype = u64
        L.orc_LF.argtypes = [vp, u64, C.c_uint8]
        L.orc_pml_batch.argtypes = [vp, vp, vp, u64, vp, vp, i32]
        L.orc_ms_batch.argtypes = [vp, vp, vp, u64, vp, vp, vp, u64, vp, i32]
        L.orc_classify_batch.argtypes = [vp, vp, u64, u64, u64, vp, vp, vp, vp]
        L.orc_pml_stats.argtypes = [vp, vp, vp, u64, vp, vp, vp]
        L.orc_max_value_thr.restype = C.c_size_t
        L.orc_max_value_thr.argtypes = [C.c_double, i32, i32, i32]
        L.orc_t1_build.restype = vp
        L.orc_t1_build.argtypes = [vp, vp, vp, u64, vp, vp, vp, vp]
        L.orc_t1_free.argtypes = [vp]
        L.orc_t1_run_of_position.restype = u64
        L.orc_t1_run_of_position.argtypes = [vp, u64]
        L.orc_t1_at.restype = C.c_uint8
        L.orc_t1_at.argtypes = [vp, u64]
        for name in ("orc_t1_rank", "orc_t1_select", "orc_t1_LF"):
            getattr(L, name).restype = u64
            getattr(L, name).argtypes = [vp, u64, C.c_uint8]
        L.orc_t1_threshold.restype = u64
        L.orc_t1_threshold.argtypes = [vp, u64]
        L.orc_t1_pml_batch.argtypes = [vp, vp, vp, u64, vp, vp, i32]
        L.orc_t1_ms_batch.argtypes = [vp, vp, vp, u64, vp, vp, i32]
        L.orc_digest_default_charhash.argtypes = [vp]
        L.orc_digest.restype = C.c_size_t
        L.orc_digest.argtypes = [i32, C.c_uint, C.c_uint, vp, vp, C.c_size_t, vp, C.c_size_t]
        L.orc_digest_batch.argtypes = [i32, C.c_uint, C.c_uint, vp, vp, vp, u64, vp, u64, vp]
        _LIB = L
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _u64(a):
    return None if a is None else np.ascontiguousarray(np.asarray(a), dtype=np.uint64)


class OracleIndex:
    """Owns an orc_index built from raw per-run arrays (numpy or torch CPU tensors)."""

    def __init__(self, heads, lens, thr, ssa=None, esa=None, doc_start=None, doc_end=None):
        def np_(x):
            if x is None:
                return None
            return x.cpu().numpy() if hasattr(x, "cpu") else np.asarray(x)

        heads = np.ascontiguousarray(np_(heads), dtype=np.uint8)
        lens, thr = _u64(np_(lens)), _u64(np_(thr))
        ssa, esa = _u64(np_(ssa)), _u64(np_(esa))
        ds, de = _u64(np_(doc_start)), _u64(np_(doc_end))
        self.r = int(heads.size)
        self.n = int(lens.sum())
        self.has_samples = ssa is not None
        self.has_docs = ds is not None
        self._h = lib().orc_build(_p(heads), _p(lens), _p(thr), self.r, _p(ssa), _p(esa), _p(ds), _p(de))
        if not self._h:
            raise RuntimeError("orc_build failed")

    @classmethod
    def from_raw(cls, raw) -> "OracleIndex":
        return cls(raw.heads, raw.lens, raw.thr, raw.ssa, raw.esa, raw.doc_start, raw.doc_end)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _LIB is not None:
            _LIB.orc_free(h)

    # primitives (for KATs)
    def run_of_position(self, p):
        return lib().orc_run_of_position(self._h, p)

    def at(self, p):
        return lib().orc_at(self._h, p)

    def rank(self, p, c):
        return lib().orc_rank(self._h, p, c)

    def select(self, i, c):
        return lib().orc_select(self._h, i, c)

    def threshold(self, k):
        return lib().orc_threshold(self._h, k)

    def LF(self, p, c):
        return lib().orc_LF(self._h, p, c)

    # batch queries
    def pml(self, seqs, offs, want_docs=False, nthreads=0):
        seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
        offs = _u64(offs)
        nreads = offs.size - 1
        out = np.zeros(max(1, int(offs[-1])), dtype=np.uint32)
        docs = np.zeros_like(out) if want_docs else None
        lib().orc_pml_batch(self._h, _p(seqs), _p(offs), nreads, _p(out), _p(docs), nthreads)
        tot = int(offs[-1])
        return (out[:tot], docs[:tot]) if want_docs else out[:tot]

    def ms(self, seqs, offs, want_docs=False, text=None, nthreads=0):
        seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
        offs = _u64(offs)
        nreads = offs.size - 1
        tot = int(offs[-1])
        ptrs = np.zeros(max(1, tot), dtype=np.uint64)
        docs = np.zeros(max(1, tot), dtype=np.uint32) if want_docs else None
        lens = None
        ntext = 0
        if text is not None:
            text = np.ascontiguousarray(text, dtype=np.uint8)
            ntext = int(text.size)
            lens = np.zeros(max(1, tot), dtype=np.uint32)
        lib().orc_ms_batch(self._h, _p(seqs), _p(offs), nreads, _p(ptrs), _p(docs), _p(text), ntext, _p(lens), nthreads)
        res = {"pointers": ptrs[:tot]}
        if want_docs:
            res["docs"] = docs[:tot]
        if lens is not None:
            res["lengths"] = lens[:tot]
        return res

    def stats(self, seqs, offs):
        seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
        offs = _u64(offs)
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        lib().orc_pml_stats(self._h, _p(seqs), _p(offs), offs.size - 1, C.byref(a), C.byref(b), C.byref(c))
        return {"steps": a.value, "jumps": b.value, "pred_jumps": c.value}


class OracleT1Index:
    """Tier T1: Elias-Fano + Huffman wavelet tree + B-run block walk (spumoni_oracle_t1.c)."""

    def __init__(self, heads, lens, thr, ssa=None, esa=None, doc_start=None, doc_end=None):
        def np_(x):
            if x is None:
                return None
            return x.cpu().numpy() if hasattr(x, "cpu") else np.asarray(x)

        heads = np.ascontiguousarray(np_(heads), dtype=np.uint8)
        lens, thr = _u64(np_(lens)), _u64(np_(thr))
        ssa, esa = _u64(np_(ssa)), _u64(np_(esa))
        ds, de = _u64(np_(doc_start)), _u64(np_(doc_end))
        self.r, self.n = int(heads.size), int(lens.sum())
        self._h = lib().orc_t1_build(_p(heads), _p(lens), _p(thr), self.r, _p(ssa), _p(esa), _p(ds), _p(de))

    @classmethod
    def from_raw(cls, raw) -> "OracleT1Index":
        return cls(raw.heads, raw.lens, raw.thr, raw.ssa, raw.esa, raw.doc_start, raw.doc_end)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _LIB is not None:
            _LIB.orc_t1_free(h)

    def run_of_position(self, p):
        return lib().orc_t1_run_of_position(self._h, p)

    def at(self, p):
        return lib().orc_t1_at(self._h, p)

    def rank(self, p, c):
        return lib().orc_t1_rank(self._h, p, c)

    def select(self, i, c):
        return lib().orc_t1_select(self._h, i, c)

    def threshold(self, k):
        return lib().orc_t1_threshold(self._h, k)

    def LF(self, p, c):
        return lib().orc_t1_LF(self._h, p, c)

    def pml(self, seqs, offs, want_docs=False, nthreads=0):
        seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
        offs = _u64(offs)
        tot = int(offs[-1])
        out = np.zeros(max(1, tot), dtype=np.uint32)
        docs = np.zeros_like(out) if want_docs else None
        lib().orc_t1_pml_batch(self._h, _p(seqs), _p(offs), offs.size - 1, _p(out), _p(docs), nthreads)
        return (out[:tot], docs[:tot]) if want_docs else out[:tot]

    def ms(self, seqs, offs, want_docs=False, nthreads=0):
        seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
        offs = _u64(offs)
        tot = int(offs[-1])
        ptrs = np.zeros(max(1, tot), dtype=np.uint64)
        docs = np.zeros(max(1, tot), dtype=np.uint32) if want_docs else None
        lib().orc_t1_ms_batch(self._h, _p(seqs), _p(offs), offs.size - 1, _p(ptrs), _p(docs), nthreads)
        res = {"pointers": ptrs[:tot]}
        if want_docs:
            res["docs"] = docs[:tot]
        return res


def classify(lengths, offs, bin_width, max_value_thr):
    lengths = np.ascontiguousarray(lengths, dtype=np.uint32)
    offs = _u64(offs)
    nreads = offs.size - 1
    found = np.zeros(nreads, dtype=np.uint8)
    above = np.zeros(nreads, dtype=np.uint32)
    below = np.zeros(nreads, dtype=np.uint32)
    ssum = np.zeros(nreads, dtype=np.uint64)
    lib().orc_classify_batch(_p(lengths), _p(offs), nreads, bin_width, max_value_thr, _p(found), _p(above), _p(below), _p(ssum))
    return found, above, below, ssum


def max_value_thr(percentile_value, is_pml, use_promotions, use_dna_letters):
    return int(lib().orc_max_value_thr(float(percentile_value), int(is_pml), int(use_promotions), int(use_dna_letters)))


DIGEST_PROMOTED, DIGEST_DNA = 1, 2


def digest_default_charhash():
    out = np.zeros(4, dtype=np.uint8)
    lib().orc_digest_default_charhash(_p(out))
    return out


def digest(kind, k, w, seq, charhash=None):
    """One read through orc_digest (src/spumoni.cpp:294-342 restated); seq: bytes / uint8 array."""
    s = np.frombuffer(bytes(seq), dtype=np.uint8) if not isinstance(seq, np.ndarray) else np.ascontiguousarray(seq, dtype=np.uint8)
    ch = None if charhash is None else np.ascontiguousarray(charhash, dtype=np.uint8)
    cap = max(1, int(k) * len(s))
    out = np.zeros(cap, dtype=np.uint8)
    n = lib().orc_digest(kind, k, w, _p(ch), _p(s), len(s), _p(out), cap)
    return out[:n].copy()


def digest_batch(kind, k, w, seqs, offs, charhash=None):
    seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
    offs = np.ascontiguousarray(offs, dtype=np.uint64)
    ch = None if charhash is None else np.ascontiguousarray(charhash, dtype=np.uint8)
    nreads = len(offs) - 1
    cap = max(1, int(k) * len(seqs))
    out = np.zeros(cap, dtype=np.uint8)
    out_offs = np.zeros(nreads + 1, dtype=np.uint64)
    lib().orc_digest_batch(kind, k, w, _p(ch), _p(seqs), _p(offs), nreads, _p(out), cap, _p(out_offs))
    return out[: int(out_offs[-1])].copy(), out_offs
