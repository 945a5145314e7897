"""ctypes binding of libspumoni_gpu.so (the C-ABI in include/spumoni_gpu.h).

This is plumbing for tests and bench: torch only provides device memory and
streams.  There is no CPU fallback -- if the library or a gfx950 device is
missing, loading / index construction raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SPUMONI_GPU_LIB: load another build of the same library (A/B experiments: tools/ab.sh)
LIB_PATH = os.environ.get("SPUMONI_GPU_LIB") or os.path.join(_HERE, "libspumoni_gpu.so")
CSRC = os.path.join(_HERE, "csrc")

SPX_MODE_PML = 0
SPX_MODE_MS = 1

EXPORTS = [
    "spx_last_error",
    "spx_device_count",
    "spx_index_from_runs",
    "spx_index_load_raw",
    "spx_index_free",
    "spx_index_stats",
    "spx_index_device_bytes",
    "spx_index_set_text",
    "spx_query_batch",
    "spx_query_batch_device",
    "spx_query_batch16",
    "spx_query_batch_device16",
    "spx_last_walk_stats",
    "spx_set_option",
    "spx_host_alloc",
    "spx_host_free",
    "spx_host_register",
    "spx_host_unregister",
    "spx_digest_capacity",
    "spx_digest_batch",
    "spx_digest_batch_device",
    "spx_digest_query_batch",
    "spx_digest_query_batch_device",
    "spx_digest_query_batch_device16",
    "spx_version",
    "spx_index_save",
    "spx_index_load_flat",
    "spx_index_clone",
    "spx_index_describe",
    "spx_last_chunk_stats",
    "spx_index_rebuild_text",
    "spx_index_copy_text",
    "spx_index_set_source_tag",
    "spx_index_source_tag",
    "spx_query_text_begin",
    "spx_query_text_fetch",
    "spx_query_text_reserve",
]
SPX_TEXT_LENGTHS, SPX_TEXT_POINTERS, SPX_TEXT_DOCS = 1, 2, 4
SPX_TEXT_UNCHECKED = 2

SPX_DIGEST_PROMOTED, SPX_DIGEST_DNA = 1, 2


class SpxClass(C.Structure):
    _fields_ = [("sum_max_bin_values", C.c_uint64), ("bins_above", C.c_uint32), ("bins_below", C.c_uint32)]


class SpxWalkStats(C.Structure):
    _fields_ = [
        ("steps", C.c_uint64),
        ("jumps", C.c_uint64),
        ("pred_jumps", C.c_uint64),
        ("row_loads", C.c_uint64),
        ("dir_loads", C.c_uint64),
        ("kernel_ms", C.c_float),
    ]


CLASS_DTYPE = np.dtype([("sum_max", "<u8"), ("above", "<u4"), ("below", "<u4")])


class SpxError(RuntimeError):
    pass


def build(force: bool = False) -> str:
    """Compile libspumoni_gpu.so for gfx950 (hipcc cross-compiles without a GPU)."""
    args = ["make", "-C", CSRC, "-j4"]
    if force:
        args.append("-B")
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return LIB_PATH


_LIB = None


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise SpxError(
                f"{LIB_PATH} is missing: build it with `make -C spumoni_amd/csrc` "
                "(there is no CPU fallback for the HIP path)"
            )
        L = C.CDLL(LIB_PATH)
        vp, u64, i32 = C.c_void_p, C.c_uint64, C.c_int
        L.spx_last_error.restype = C.c_char_p
        L.spx_device_count.restype = i32
        L.spx_index_from_runs.restype = vp
        L.spx_index_from_runs.argtypes = [vp, vp, vp, u64, vp, vp, vp, vp, i32, i32]
        L.spx_index_load_raw.restype = vp
        L.spx_index_load_raw.argtypes = [C.c_char_p, i32, i32]
        L.spx_index_free.argtypes = [vp]
        L.spx_index_stats.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]
        L.spx_index_device_bytes.argtypes = [vp, C.POINTER(u64)]
        L.spx_index_set_text.argtypes = [vp, vp, u64, i32]
        L.spx_index_set_source_tag.argtypes = [vp, C.c_char_p]
        L.spx_index_source_tag.restype = C.c_char_p
        L.spx_index_source_tag.argtypes = [vp]
        L.spx_query_text_begin.argtypes = [vp, i32, i32, C.c_uint32, C.c_uint32, vp, vp, u64, vp, C.c_uint32, vp, u64, u64, vp]
        L.spx_query_text_fetch.argtypes = [vp, vp, vp]
        L.spx_query_text_reserve.argtypes = [vp, i32, i32, C.c_uint32, u64, u64, C.c_uint32, i32, vp]
        L.spx_query_batch.argtypes = [vp, i32, vp, vp, u64, vp, vp, vp, vp, u64, u64]
        L.spx_query_batch_device.argtypes = [vp, i32, vp, vp, u64, u64, vp, vp, vp, vp, u64, u64, vp]
        L.spx_query_batch16.argtypes = [vp, i32, vp, vp, u64, vp, vp, vp, vp, u64, u64]
        L.spx_query_batch_device16.argtypes = [vp, i32, vp, vp, u64, u64, vp, vp, vp, vp, u64, u64, vp]
        L.spx_last_walk_stats.argtypes = [vp, C.POINTER(SpxWalkStats)]
        L.spx_set_option.argtypes = [vp, C.c_char_p, C.c_int64]
        L.spx_host_alloc.restype = vp
        L.spx_host_alloc.argtypes = [C.c_size_t]
        L.spx_host_free.argtypes = [vp]
        L.spx_host_register.restype = i32
        L.spx_host_register.argtypes = [vp, C.c_size_t]
        L.spx_host_unregister.restype = i32
        L.spx_host_unregister.argtypes = [vp]
        u32 = C.c_uint32
        L.spx_digest_capacity.restype = u64
        L.spx_digest_capacity.argtypes = [i32, u32, u64]
        L.spx_digest_batch.argtypes = [vp, i32, u32, u32, vp, vp, u64, vp, u64, vp]
        L.spx_digest_batch_device.argtypes = [vp, i32, u32, u32, vp, vp, u64, u64, vp, u64, vp, vp]
        L.spx_digest_query_batch.argtypes = [vp, i32, i32, u32, u32, vp, vp, u64, vp, u64, vp, vp, vp, vp, u64, u64]
        for fn in (L.spx_digest_query_batch_device, L.spx_digest_query_batch_device16):
            fn.argtypes = [vp, i32, i32, u32, u32, vp, vp, u64, u64, vp, u64, vp, vp, vp, vp, vp, u64, u64, vp]
        L.spx_version.restype = C.c_char_p
        L.spx_index_save.argtypes = [vp, C.c_char_p]
        L.spx_index_load_flat.restype = vp
        L.spx_index_load_flat.argtypes = [C.c_char_p, i32]
        L.spx_index_clone.restype = vp
        L.spx_index_clone.argtypes = [vp, i32]
        L.spx_index_describe.argtypes = [vp, C.c_char_p, C.c_size_t]
        L.spx_last_chunk_stats.argtypes = [vp, C.POINTER(u64 * 4)]
        L.spx_index_rebuild_text.argtypes = [vp]
        L.spx_index_copy_text.argtypes = [vp, vp, u64, i32, C.POINTER(u64)]
        _LIB = L
    return _LIB


def version() -> str:
    """Layout + kernel version of the loaded library (keys the .spx cache and the measured traffic)."""
    return lib().spx_version().decode()


def _check(rc: int):
    if rc != 0:
        raise SpxError(f"spx error {rc}: {lib().spx_last_error().decode()}")


def _np_ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _t_ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class Index:
    """An spx_index on one GPU."""

    def __init__(self, handle, device: int):
        self._h = handle
        self.device = device
        n, r = C.c_uint64(), C.c_uint64()
        _check(lib().spx_index_stats(self._h, C.byref(n), C.byref(r)))
        self.n, self.r = n.value, r.value

    # -- construction -----------------------------------------------------
    @classmethod
    def from_raw(cls, raw, device: int = 0) -> "Index":
        """raw: spumoni_amd.synth.RawIndex (torch tensors on CPU or on `device`)."""
        import torch

        on_dev = raw.heads.is_cuda
        keep = []

        def prep(t, dt):
            if t is None:
                return None
            t = t.to(dt).contiguous()
            keep.append(t)
            return t

        heads = prep(raw.heads, torch.uint8)
        # int64 and uint64 share the bit pattern for the value range used here
        lens, thr = prep(raw.lens, torch.int64), prep(raw.thr, torch.int64)
        ssa, esa = prep(raw.ssa, torch.int64), prep(raw.esa, torch.int64)
        ds, de = prep(raw.doc_start, torch.int64), prep(raw.doc_end, torch.int64)
        if on_dev:
            torch.cuda.synchronize()
        h = lib().spx_index_from_runs(
            _t_ptr(heads), _t_ptr(lens), _t_ptr(thr), raw.r, _t_ptr(ssa), _t_ptr(esa), _t_ptr(ds), _t_ptr(de),
            1 if on_dev else 0, device,
        )
        if not h:
            raise SpxError(lib().spx_last_error().decode())
        ix = cls(h, device)
        if raw.text is not None:
            ix.set_text(raw.text)
        return ix

    @classmethod
    def load_raw(cls, prefix: str, mode: int = SPX_MODE_PML, device: int = 0) -> "Index":
        h = lib().spx_index_load_raw(prefix.encode(), mode, device)
        if not h:
            raise SpxError(lib().spx_last_error().decode())
        return cls(h, device)

    @classmethod
    def load_flat(cls, path: str, device: int = 0) -> "Index":
        """An index from a flat-layout cache written by save()."""
        h = lib().spx_index_load_flat(path.encode(), device)
        if not h:
            raise SpxError(lib().spx_last_error().decode())
        return cls(h, device)

    def save(self, path: str) -> None:
        _check(lib().spx_index_save(self._h, path.encode()))

    def set_source_tag(self, tag: str) -> None:
        """The caller's fingerprint of what the index was built from; travels with save() / load_flat() / clone()."""
        _check(lib().spx_index_set_source_tag(self._h, tag.encode()))

    def source_tag(self) -> str:
        return lib().spx_index_source_tag(self._h).decode()

    def clone(self, device: int) -> "Index":
        """A copy of this index on `device` (device-to-device, no re-flattening); onto the SAME device: a second query context --
        scratch, counters and stream of its own -- over the same arrays (nothing is copied)."""
        h = lib().spx_index_clone(self._h, device)
        if not h:
            raise SpxError(lib().spx_last_error().decode())
        return Index(h, device)

    def describe(self) -> dict:
        import json

        buf = C.create_string_buffer(1024)
        _check(lib().spx_index_describe(self._h, buf, 1024))
        return json.loads(buf.value.decode())

    def rebuild_text(self) -> None:
        """The indexed text from the MS index itself (LF chains from the SA samples): no text file needed."""
        _check(lib().spx_index_rebuild_text(self._h))

    def text(self) -> np.ndarray:
        """The text the index holds (set or rebuilt), as a host array."""
        n = C.c_uint64()
        _check(lib().spx_index_copy_text(self._h, None, 0, 0, C.byref(n)))
        out = np.zeros(n.value, dtype=np.uint8)
        _check(lib().spx_index_copy_text(self._h, _np_ptr(out), n.value, 0, C.byref(n)))
        return out

    def set_text(self, text, unchecked: bool = False) -> None:
        """unchecked: skip the text-against-index validation (synthetic indexes that are no text's BWT)."""
        t = text.contiguous()
        where = (1 if t.is_cuda else 0) | (SPX_TEXT_UNCHECKED if unchecked else 0)
        _check(lib().spx_index_set_text(self._h, _t_ptr(t), t.numel(), where))

    def close(self):
        h, self._h = self._h, None
        if h:
            lib().spx_index_free(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def device_bytes(self) -> int:
        b = C.c_uint64()
        _check(lib().spx_index_device_bytes(self._h, C.byref(b)))
        return b.value

    def set_option(self, key: str, value: int) -> None:
        _check(lib().spx_set_option(self._h, key.encode(), value))

    # -- queries, host buffers (numpy) --------------------------------------
    def query_host(self, mode, seqs, offs, want_lengths=True, want_docs=False, classify=None, bits=32):
        """bits=16: the 16-bit entry point (uint16 lengths / docs; reads shorter than 65536)."""
        seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        nreads = offs.size - 1
        tot = int(offs[-1]) if nreads > 0 else 0
        vt = np.uint16 if bits == 16 else np.uint32
        lens = np.zeros(max(tot, 1) + 8, dtype=vt) if want_lengths else None  # PML: None + classify = report only
        ptrs = np.zeros(max(tot, 1), dtype=np.uint64) if mode == SPX_MODE_MS else None
        docs = np.zeros(max(tot, 1) + 8, dtype=vt) if want_docs else None
        cls_ = np.zeros(max(nreads, 1), dtype=CLASS_DTYPE) if classify else None
        bw, thr = classify if classify else (0, 0)
        fn = lib().spx_query_batch16 if bits == 16 else lib().spx_query_batch
        _check(fn(self._h, mode, _np_ptr(seqs), _np_ptr(offs), nreads, _np_ptr(lens),
                  _np_ptr(ptrs), _np_ptr(docs), _np_ptr(cls_), bw, thr))
        out = {}
        if lens is not None:
            out["lengths"] = lens[:tot]
        if ptrs is not None:
            out["pointers"] = ptrs[:tot]
        if docs is not None:
            out["docs"] = docs[:tot]
        if cls_ is not None:
            out["class"] = cls_[:nreads]
        return out

    def query_text(self, mode, seqs, offs, gap=None, streams=SPX_TEXT_LENGTHS, digest=(0, 0, 0), classify=None):
        """The vectors as the text of the output files (spx_query_text_begin / _fetch).  Returns
        {"text": [bytes | None] * 3, "line_start": [uint64 array | None] * 3, "class": ...}; stream i = lengths,
        pointers, document ids."""
        seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        nreads = offs.size - 1
        g = None if gap is None else np.ascontiguousarray(gap, dtype=np.uint32)
        cls_ = np.zeros(max(nreads, 1), dtype=CLASS_DTYPE) if classify else None
        bw, thr = classify if classify else (0, 0)
        nbytes = (C.c_uint64 * 3)()
        _check(lib().spx_query_text_begin(self._h, mode, digest[0], digest[1], digest[2], _np_ptr(seqs), _np_ptr(offs), nreads,
                                          _np_ptr(g), streams, _np_ptr(cls_), bw, thr, C.cast(nbytes, C.c_void_p)))
        bufs = [np.zeros(int(nbytes[i]) + 1, dtype=np.uint8) if nbytes[i] else None for i in range(3)]
        starts = [np.zeros(nreads + 1, dtype=np.uint64) if nbytes[i] else None for i in range(3)]
        tp = (C.c_void_p * 3)(*[_np_ptr(b) for b in bufs])
        lp = (C.c_void_p * 3)(*[_np_ptr(b) for b in starts])
        _check(lib().spx_query_text_fetch(self._h, C.cast(tp, C.c_void_p), C.cast(lp, C.c_void_p)))
        return {"text": [None if b is None else b[: int(nbytes[i])].tobytes() for i, b in enumerate(bufs)],
                "line_start": starts, "class": None if cls_ is None else cls_[:nreads]}

    def reserve_text(self, mode, max_chars, max_reads, streams=SPX_TEXT_LENGTHS, classify=False, digest=(0, 0), text_bytes=None):
        """spx_query_text_reserve: the device scratch of query_text calls up to this size, allocated now (a hint)."""
        tb = None
        if text_bytes is not None:
            tb = (C.c_uint64 * 3)(*[int(x) for x in text_bytes])
        _check(lib().spx_query_text_reserve(self._h, mode, digest[0], digest[1], int(max_chars), int(max_reads), streams,
                                            1 if classify else 0, None if tb is None else C.cast(tb, C.c_void_p)))

    # -- queries, device buffers (torch tensors on self.device) -------------
    def query_device(self, mode, d_seqs, d_offs, total_chars, d_lengths=None, d_pointers=None, d_docs=None,
                     d_class=None, bin_width=0, max_value_thr=0, stream=None):
        """Enqueue on `stream` (a torch.cuda.Stream or None = torch's current stream).  int16 / uint16
        tensors for d_lengths / d_docs select the 16-bit entry point."""
        import torch

        st = stream if stream is not None else torch.cuda.current_stream(self.device)
        nreads = d_offs.numel() - 1
        narrow = any(t is not None and t.element_size() == 2 for t in (d_lengths, d_docs))
        fn = lib().spx_query_batch_device16 if narrow else lib().spx_query_batch_device
        _check(fn(
            self._h, mode, _t_ptr(d_seqs), _t_ptr(d_offs), nreads, total_chars, _t_ptr(d_lengths),
            _t_ptr(d_pointers), _t_ptr(d_docs), _t_ptr(d_class), bin_width, max_value_thr,
            C.c_void_p(st.cuda_stream)))

    # -- minimizer digestion (run -m / -a) ------------------------------------
    def digest_host(self, kind, k, w, seqs, offs):
        """(digested seqs, offsets) of a batch of upper-cased reads; host buffers."""
        seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        nreads = offs.size - 1
        tot = int(offs[-1]) if nreads > 0 else 0
        cap = int(lib().spx_digest_capacity(kind, k, tot))
        out = np.zeros(cap, dtype=np.uint8)
        out_offs = np.zeros(nreads + 1, dtype=np.uint64)
        _check(lib().spx_digest_batch(self._h, kind, k, w, _np_ptr(seqs), _np_ptr(offs), nreads, _np_ptr(out), cap,
                                      _np_ptr(out_offs)))
        return out[: int(out_offs[-1])], out_offs

    def digest_device(self, kind, k, w, d_seqs, d_offs, total_chars, stream=None):
        """Digest on the device; returns (d_out_seqs, d_out_offs) torch tensors that can go
        straight into query_device (total = d_out_offs[-1])."""
        import torch

        st = stream if stream is not None else torch.cuda.current_stream(self.device)
        nreads = d_offs.numel() - 1
        cap = int(lib().spx_digest_capacity(kind, k, total_chars))
        d_out = torch.empty(cap, dtype=torch.uint8, device=d_seqs.device)
        d_out_offs = torch.empty(nreads + 1, dtype=torch.int64, device=d_seqs.device)
        _check(lib().spx_digest_batch_device(self._h, kind, k, w, _t_ptr(d_seqs), _t_ptr(d_offs), nreads, total_chars,
                                             _t_ptr(d_out), cap, _t_ptr(d_out_offs), C.c_void_p(st.cuda_stream)))
        return d_out, d_out_offs

    def digest_query_device(self, mode, kind, k, w, d_seqs, d_offs, total_chars, d_lengths=None, d_pointers=None, d_docs=None,
                            d_class=None, bin_width=0, max_value_thr=0, stream=None, work=None):
        """Digest and query in one call, everything on the device (spx_digest_query_batch_device[16]): returns
        (d_out_offs, work).  The outputs (sized for total_chars entries) are laid out at d_out_offs; `work` = (digested
        bytes, offsets) buffers that a caller may hand back to avoid the allocations."""
        import torch

        st = stream if stream is not None else torch.cuda.current_stream(self.device)
        nreads = d_offs.numel() - 1
        cap = int(lib().spx_digest_capacity(kind, k, total_chars))
        if work is None or work[0].numel() < cap or work[1].numel() < nreads + 1:
            work = (torch.empty(cap, dtype=torch.uint8, device=d_seqs.device),
                    torch.empty(nreads + 1, dtype=torch.int64, device=d_seqs.device))
        narrow = any(t is not None and t.element_size() == 2 for t in (d_lengths, d_docs))
        fn = lib().spx_digest_query_batch_device16 if narrow else lib().spx_digest_query_batch_device
        _check(fn(self._h, mode, kind, k, w, _t_ptr(d_seqs), _t_ptr(d_offs), nreads, total_chars, _t_ptr(work[0]), work[0].numel(),
                  _t_ptr(work[1]), _t_ptr(d_lengths), _t_ptr(d_pointers), _t_ptr(d_docs), _t_ptr(d_class), bin_width, max_value_thr,
                  C.c_void_p(st.cuda_stream)))
        return work[1], work

    def digest_query_host(self, mode, kind, k, w, seqs, offs, want_lengths=True, want_docs=False, classify=None):
        """digest + query in one call (the reference's per-read loop body, for a batch)."""
        seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        nreads = offs.size - 1
        tot = int(offs[-1]) if nreads > 0 else 0
        cap = int(lib().spx_digest_capacity(kind, k, tot))
        out_offs = np.zeros(nreads + 1, dtype=np.uint64)
        lens = np.zeros(cap, dtype=np.uint32) if want_lengths else None
        ptrs = np.zeros(cap, dtype=np.uint64) if mode == SPX_MODE_MS else None
        docs = np.zeros(cap, dtype=np.uint32) if want_docs else None
        cls_ = np.zeros(max(nreads, 1), dtype=CLASS_DTYPE) if classify else None
        bw, thr = classify if classify else (0, 0)
        _check(lib().spx_digest_query_batch(self._h, mode, kind, k, w, _np_ptr(seqs), _np_ptr(offs), nreads,
                                            _np_ptr(out_offs), cap, _np_ptr(lens), _np_ptr(ptrs), _np_ptr(docs),
                                            _np_ptr(cls_), bw, thr))
        dt = int(out_offs[-1])
        out = {"offsets": out_offs}
        if lens is not None:
            out["lengths"] = lens[:dt]
        if ptrs is not None:
            out["pointers"] = ptrs[:dt]
        if docs is not None:
            out["docs"] = docs[:dt]
        if cls_ is not None:
            out["class"] = cls_[:nreads]
        return out

    def last_chunk_stats(self) -> dict:
        """Chunked walk of the last query: chunk size (0 = it ran the plain walk), characters walked a second
        time to join the chunks, reads that fell back to the plain walk."""
        o = (C.c_uint64 * 4)()
        _check(lib().spx_last_chunk_stats(self._h, C.byref(o)))
        return {"chunk_len": o[0], "chunks_bound": o[1], "rewalked_chars": o[2], "fallback_reads": o[3]}

    def last_stats(self) -> dict:
        s = SpxWalkStats()
        _check(lib().spx_last_walk_stats(self._h, C.byref(s)))
        return {f[0]: getattr(s, f[0]) for f in SpxWalkStats._fields_}


def digester(device: int = 0) -> "Index":
    """A minimal index handle, for callers that only want the device digestion (index builders)."""
    import torch

    from .synth import RawIndex

    z = torch.zeros(1, dtype=torch.int64)
    return Index.from_raw(RawIndex(heads=torch.zeros(1, dtype=torch.uint8), lens=torch.ones(1, dtype=torch.int64),
                                   thr=z, n=1), device)


def pinned_array(shape, dtype):
    """numpy array backed by spx_host_alloc (page-locked) memory; keep the returned owner alive."""
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    p = lib().spx_host_alloc(max(n, 1))
    if not p:
        raise SpxError(lib().spx_last_error().decode())
    buf = (C.c_char * max(n, 1)).from_address(p)
    arr = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    free = lib().spx_host_free

    class _Owner:
        def __del__(self_inner):
            try:
                free(p)
            except Exception:
                pass

    return arr, _Owner()


def pad_seqs(seqs):
    """Return a copy of seqs with the slack spx_query_batch_device asks for
    (readable for round_up(n, 4) + 32 bytes)."""
    import torch

    n = seqs.numel()
    padded = torch.zeros(((n + 3) // 4) * 4 + 32, dtype=torch.uint8, device=seqs.device)
    padded[:n] = seqs
    return padded
