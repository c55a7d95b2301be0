"""fmindex-collection_amd — MI355X-native backward-search engine (host-side Python mirror over the C-ABI).

The product is libfmgpu.so (hand-written HIP for gfx950, include/fmgpu.h); this package is the thin Python
host layer used by the tests and bench.py.  It mirrors the reference's names for the hot path:
FMIndex / BiFMIndex (fmindex/FMIndex.h, fmindex/BiFMIndex.h), search_no_errors.search (search/SearchNoErrors.h),
search_backtracking.search (search/Backtracking.h), search_ng26.search (search/SearchNg26.h), LocateLinear
(locate.h), search_scheme.* (search_scheme/).  The C++ mirror of the template API is include/fmc_gpu.hpp.

There is no CPU fallback anywhere in this package: without libfmgpu.so and a GPU every compute call raises.
"""
import ctypes as C
import os

import numpy as np

from . import capi
from .capi import FmgpuError, DeviceBuffer, LAYOUTS, UINT64_MAX, HIT_DTYPE
from . import search_scheme  # noqa: F401

__all__ = ["FMIndex", "BiFMIndex", "search_no_errors", "search_backtracking", "search_ng26", "search_ng21", "search", "search_n", "search_best", "LocateLinear",
           "search_scheme", "FmgpuError", "DeviceBuffer", "flatten", "device_count", "Replicas", "options"]


class _Options:
    """the library's process-wide options (fmgpu_set_option, include/fmgpu.h): `options["pair_table"] = 0`, `del options["pair_table"]` puts the default back,
    `with options(pair_table=0, kernel_select=capi.SEL_GENERAL_DFS): ...` sets and restores"""

    def __setitem__(self, name, value):
        capi.set_option(name, value)

    def __getitem__(self, name):
        return capi.get_option(name)

    def __delitem__(self, name):
        capi.set_option(name, capi.OPTION_DEFAULTS[name])

    def pop(self, name, default=None):
        del self[name]

    def __call__(self, **kw):
        import contextlib

        @contextlib.contextmanager
        def scope():
            old = {k: self[k] for k in kw}
            try:
                for k, v in kw.items():
                    self[k] = v
                yield self
            finally:
                for k, v in old.items():
                    self[k] = v
        return scope()


options = _Options()


def device_count():
    n = C.c_int()
    capi.check(capi.lib().fmgpu_device_count(C.byref(n)))
    return n.value


def flatten(sequences):
    """Sequences (list of byte sequences) -> (qbuf uint8[total], qoff uint64[nq+1]) — the ABI's query format"""
    lens = np.fromiter((len(q) for q in sequences), dtype=np.uint64, count=len(sequences))
    qoff = np.zeros(len(sequences) + 1, dtype=np.uint64)
    np.cumsum(lens, out=qoff[1:])
    total = int(qoff[-1])
    qbuf = np.zeros(max(total, 1), dtype=np.uint8)
    if total:
        qbuf[:total] = np.concatenate([np.asarray(q, dtype=np.uint8) for q in sequences if len(q)])
    return qbuf, qoff


def _u64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.uint64))


class _StringArrays:
    """host arrays of one reference String object, kept alive while the descriptor is in use"""

    def __init__(self, layout, sigma, n, blocks=None, super_blocks=None, nodes=None, levels=None, super_row=None):
        self.layout, self.sigma, self.n = layout, sigma, n
        self.super_row = super_row or sigma                      # entries per super-block row (FlattenedBitvectors2L::l0 has sigma + 1)
        # EPRV3/4/5/7: `blocks` = the bits array, `levels` = counter arrays bottom-up (None where the layout has none)
        self.levels = None if levels is None else [None if a is None else np.ascontiguousarray(a).view(np.uint8) for a in levels]
        self.blocks = None if blocks is None else np.ascontiguousarray(blocks).view(np.uint8)
        self.super_blocks = None if super_blocks is None else _u64(super_blocks).reshape(-1)
        self.nodes = nodes  # list of (superblocks u64, blocks u8, bits u64, total_length)
        self._node_arr = None

    def desc(self):
        d = capi.StringDesc()
        d.layout = LAYOUTS[self.layout]
        d.sigma = self.sigma
        d.n = self.n
        if self.blocks is not None:
            d.blocks = self.blocks.ctypes.data
            d.blocks_bytes = self.blocks.nbytes
            d.super_blocks = self.super_blocks.ctypes.data_as(capi.u64p)
            d.n_super_blocks = self.super_blocks.size // self.super_row
        if self.levels is not None:
            for k, a in enumerate(self.levels[:3]):
                if a is not None and a.size:
                    d.levels[k] = a.ctypes.data
                    d.level_bytes[k] = a.nbytes
        if self.nodes is not None:
            arr = (capi.WaveletNode * len(self.nodes))()
            keep = []
            for k, (sb, bl, bits, total) in enumerate(self.nodes):
                sb, bl, bits = _u64(sb), np.ascontiguousarray(bl, dtype=np.uint8), _u64(bits)
                keep.append((sb, bl, bits))
                arr[k].superblocks = sb.ctypes.data_as(capi.u64p); arr[k].n_superblocks = sb.size
                arr[k].blocks = bl.ctypes.data_as(capi.u8p); arr[k].n_blocks = bl.size
                arr[k].bits = bits.ctypes.data_as(capi.u64p); arr[k].n_bits = bits.size
                arr[k].total_length = int(total)
            self._node_arr, self._keep = arr, keep
            d.nodes = arr
            d.n_nodes = len(self.nodes)
        return d


class FMIndex:
    """fmindex/FMIndex.h:14-134 (bidirectional=False) / fmindex/BiFMIndex.h:17-216 (BiFMIndex subclass), resident in HBM.

    Construct from the arrays a reference index object holds (`from_reference_arrays`) or from sequences with the
    GPU builder (`from_sequences`, replaces the libsais-based constructor fmindex/FMIndex.h:58-104)."""

    bidirectional = False

    def __init__(self, handle, keep=None, built=None):
        self._h = handle
        self._keep = keep
        self._built = built
        n, sigma, layout, bidir, dbytes = C.c_uint64(), C.c_int32(), C.c_int32(), C.c_int32(), C.c_uint64()
        capi.check(capi.lib().fmgpu_index_info(self._h, C.byref(n), C.byref(sigma), C.byref(layout), C.byref(bidir), C.byref(dbytes)))
        self.n, self.Sigma, self.layout, self.device_bytes = n.value, sigma.value, capi.LAYOUT_NAMES[layout.value], dbytes.value
        self.bidirectional = bool(bidir.value)

    # -------------------------------------------------------------- construction
    @classmethod
    def from_reference_arrays(cls, bwt, C_array, bwt_rev=None, sparse=None):
        """bwt / bwt_rev: dict(layout, sigma, n, blocks, super_blocks | nodes); sparse: dict(n, l0, l1, bits, fields[2])"""
        sa = _StringArrays(**bwt)
        desc = capi.IndexDesc()
        desc.bwt = sa.desc()
        keep = [sa]
        if bwt_rev is not None:
            sr = _StringArrays(**bwt_rev)
            rdesc = sr.desc()
            desc.bwt_rev = C.pointer(rdesc)
            keep += [sr, rdesc]
        Carr = _u64(C_array)
        desc.C = Carr.ctypes.data_as(capi.u64p)
        keep.append(Carr)
        if sparse is not None:
            sd = capi.SparseArrayDesc()
            l0, l1, bits = _u64(sparse["l0"]), np.ascontiguousarray(sparse["l1"], dtype=np.uint16), _u64(sparse["bits"])
            sd.n = sparse["n"]
            sd.l0 = l0.ctypes.data_as(capi.u64p); sd.n_l0 = l0.size
            sd.l1 = l1.ctypes.data_as(capi.u16p); sd.n_l1 = l1.size
            sd.bits = bits.ctypes.data_as(capi.u64p); sd.n_bit_words = bits.size
            keep += [l0, l1, bits]
            for f in range(2):
                fd = sparse["fields"][f]
                data = _u64(fd["data"])
                keep.append(data)
                sd.field[f].data = data.ctypes.data_as(capi.u64p); sd.field[f].n_words = data.size
                sd.field[f].bit_count = int(fd["bitCount"]); sd.field[f].bits = int(fd["bits"])
                sd.field[f].largest_value = int(fd["largestValue"]); sd.field[f].common_divisor = int(fd["commonDivisor"])
            desc.annotated_array = C.pointer(sd)
            keep.append(sd)
        h = C.c_void_p()
        capi.check(capi.lib().fmgpu_index_create(C.byref(desc), C.byref(h)))
        return cls(h, keep=None)   # the library copied everything; host arrays may go

    @classmethod
    def from_sequences(cls, sequences, sigma, layout="IB16", sampling_rate=16, bidirectional=None, keep_host=False):
        """GPU construction from Sequences (list of rank sequences, or (qbuf, qoff) already flattened / resident in HBM)"""
        if bidirectional is None:
            bidirectional = cls.bidirectional
        if isinstance(sequences, tuple):
            sbuf, soff = sequences
            nseq = (soff.nbytes // 8 if not hasattr(soff, "__len__") else len(soff)) - 1
        else:
            sbuf, soff = flatten(sequences)
            nseq = len(sequences)
        h, b = C.c_void_p(), C.c_void_p()
        capi.check(capi.lib().fmgpu_build_index(capi.ptr(sbuf), capi.ptr(soff), nseq, sigma, LAYOUTS[layout], sampling_rate,
                                                1 if bidirectional else 0, 1 if keep_host else 0, C.byref(h),
                                                C.byref(b) if keep_host else None))
        return cls(h, built=b if keep_host else None)

    # -------------------------------------------------------------- index file (replaces saveIndex / loadIndex, fmindex/diskStorage.h:12-27)
    def save(self, path, tables=True):
        """write the index to this library's own flat file (header + every device array with a checksum); tables=True keeps the optional tables the
        handle holds right now.  Not the reference's cereal format (include/fmgpu.h)"""
        capi.check(capi.lib().fmgpu_index_save(self._h, os.fsencode(path), 1 if tables else 0))
        return self

    @classmethod
    def load(cls, path):
        """an index written by save(); FMIndex.load returns a BiFMIndex object for a bidirectional file (and vice versa): the file says what it holds"""
        h = C.c_void_p()
        capi.check(capi.lib().fmgpu_index_load(os.fsencode(path), C.byref(h)))
        x = FMIndex(h)
        if x.bidirectional:
            x.__class__ = BiFMIndex
        return x

    def clone(self):
        """a copy of the handle on the calling thread's current device, made device to device (fmgpu_index_clone): every array incl. the optional tables"""
        h = C.c_void_p()
        capi.check(capi.lib().fmgpu_index_clone(self._h, C.byref(h)))
        x = FMIndex(h)
        if x.bidirectional:
            x.__class__ = BiFMIndex
        return x

    def built_array(self, part, dtype=np.uint8):
        """host copy of a construction by-product (keep_host=True): 0 = BWT bytes, 1 = BWT of the reversed text, 2 = C"""
        if not self._built:
            raise ValueError("index was not built with keep_host=True")
        p, nb = C.c_void_p(), C.c_uint64()
        capi.check(capi.lib().fmgpu_built_get(self._built, part, C.byref(p), C.byref(nb)))
        if nb.value == 0:
            return np.zeros(0, dtype=dtype)
        buf = (C.c_uint8 * nb.value).from_address(p.value)
        return np.frombuffer(buf, dtype=dtype)

    def close(self):
        if getattr(self, "_built", None):
            capi.lib().fmgpu_built_free(self._built)
            self._built = None
        if getattr(self, "_h", None):
            capi.lib().fmgpu_index_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def size(self):
        return self.n

    @property
    def row_bits(self):
        """32 or 64: the width of the device tables the index is held in (include/fmgpu.h, "row width")"""
        b = C.c_int32()
        capi.check(capi.lib().fmgpu_index_row_bits(self._h, C.byref(b)))
        return b.value

    @property
    def formats(self):
        """FMT_* bits (capi): what the bwt is held in at this moment — which kernel serves a search"""
        m = C.c_uint32()
        capi.check(capi.lib().fmgpu_index_formats(self._h, C.byref(m)))
        return m.value

    def _refresh_bytes(self):
        dbytes = C.c_uint64()
        capi.check(capi.lib().fmgpu_index_info(self._h, None, None, None, None, C.byref(dbytes)))
        self.device_bytes = dbytes.value
        return self

    def accelerate_lf(self, enable=True):
        """build / drop the explicit LF tables (one word per row and direction); without them the index is the bit-packed occurrence table alone"""
        capi.check(capi.lib().fmgpu_index_accelerate_lf(self._h, 1 if enable else 0))
        return self._refresh_bytes()

    # -------------------------------------------------------------- cursor steps (fmindex/FMIndexCursor.h:33-53, fmindex/BiFMIndexCursor.h:58-128)
    def extend(self, lb, lb_rev, length, symb=None, right=False):
        """extendLeft / extendRight of a batch of cursors.  symb = one symbol per cursor -> (lb, lb_rev, len) arrays of the same shape;
        symb = None -> extendLeft() / extendRight() over all symbols: arrays of shape (count, Sigma)"""
        lb, length = _u64(lb).reshape(-1), _u64(length).reshape(-1)
        rev = _u64(lb_rev).reshape(-1) if lb_rev is not None else None
        fan = 1 if symb is not None else self.Sigma
        sy = None if symb is None else np.ascontiguousarray(np.broadcast_to(np.asarray(symb, dtype=np.uint8), lb.shape))
        olb, olen = np.empty(lb.size * fan, dtype=np.uint64), np.empty(lb.size * fan, dtype=np.uint64)
        orev = np.empty(lb.size * fan, dtype=np.uint64) if rev is not None else None
        capi.check(capi.lib().fmgpu_cursor_extend(self._h, 1 if right else 0, lb.size, capi.ptr(lb), capi.ptr(rev), capi.ptr(length), capi.ptr(sy),
                                                  capi.ptr(olb), capi.ptr(orev), capi.ptr(olen), None))
        if fan > 1:
            olb, olen = olb.reshape(-1, fan), olen.reshape(-1, fan)
            orev = None if orev is None else orev.reshape(-1, fan)
        return olb, orev, olen

    def accelerate_search(self, prefix_len=11, walk=True):
        """BiFMIndex: prefix table for the exact first part of a search + walk tables (walk: True / 1 = LF, LF^2, LF^3 per row; 2 = LF^16 with
        the 16 symbols met; 3 = both); results are unchanged"""
        capi.check(capi.lib().fmgpu_index_accelerate_search(self._h, prefix_len, int(walk)))
        dbytes = C.c_uint64()
        capi.check(capi.lib().fmgpu_index_info(self._h, None, None, None, None, C.byref(dbytes)))
        self.device_bytes = dbytes.value
        return self

    def accelerate_locate(self, enable=True):
        """keep the (seqId, pos, steps) answer of every row (12 bytes per row): locate becomes one load per row; results are unchanged"""
        capi.check(capi.lib().fmgpu_index_accelerate_locate(self._h, 1 if enable else 0))
        dbytes = C.c_uint64()
        capi.check(capi.lib().fmgpu_index_info(self._h, None, None, None, None, C.byref(dbytes)))
        self.device_bytes = dbytes.value
        return self

    def accelerate(self, kstep=3, lut_len=0, walk=False):
        """add (kstep >= 2) or drop (0) the multi-symbol-step table used by exact search; lut_len > 0 adds the table of the
        intervals of all strings of that many symbols, walk=True the per-row LF^J + symbols table; results are unchanged"""
        capi.check(capi.lib().fmgpu_index_accelerate_exact(self._h, kstep, lut_len, int(walk)))      # walk: False / True (LF^J) / 2 (LF^J and LF^2J)
        n, sigma, layout, bidir, dbytes = C.c_uint64(), C.c_int32(), C.c_int32(), C.c_int32(), C.c_uint64()
        capi.check(capi.lib().fmgpu_index_info(self._h, C.byref(n), C.byref(sigma), C.byref(layout), C.byref(bidir), C.byref(dbytes)))
        self.device_bytes = dbytes.value
        return self

    # -------------------------------------------------------------- String_c batch (string/concepts.h:25-87)
    def _string_query(self, which, idx, symb, what):
        idx = _u64(idx)
        symb = np.ascontiguousarray(np.broadcast_to(np.asarray(symb, dtype=np.uint8), idx.shape))
        what = np.ascontiguousarray(np.broadcast_to(np.asarray(what, dtype=np.uint8), idx.shape))
        out = np.empty(idx.shape, dtype=np.uint64)
        capi.check(capi.lib().fmgpu_string_query(self._h, which, capi.ptr(idx), capi.ptr(symb), capi.ptr(what), idx.size,
                                                 capi.ptr(out), None))
        return out

    def rank(self, idx, symb, rev=False):
        return self._string_query(1 if rev else 0, idx, symb, 0)

    def prefix_rank(self, idx, symb, rev=False):
        return self._string_query(1 if rev else 0, idx, symb, 1)

    def symbol(self, idx, rev=False):
        return self._string_query(1 if rev else 0, idx, 0, 2)

    # -------------------------------------------------------------- locate (fmindex/FMIndex.h:113-124)
    def locate(self, rows, want_stats=False):
        rows = _u64(rows)
        seq, pos, steps = (np.empty(rows.shape, dtype=np.uint64) for _ in range(3))
        st = capi.Stats()
        capi.check(capi.lib().fmgpu_locate(self._h, capi.ptr(rows), rows.size, capi.ptr(seq), capi.ptr(pos), capi.ptr(steps),
                                           C.byref(st) if want_stats else None, None))
        return (seq, pos, steps, st) if want_stats else (seq, pos, steps)


class BiFMIndex(FMIndex):
    bidirectional = True


def _queries(queries):
    if isinstance(queries, tuple):
        qbuf, qoff = queries
        nq = (qoff.nbytes // 8 if not hasattr(qoff, "__len__") else len(qoff)) - 1
        return qbuf, qoff, nq
    qbuf, qoff = flatten(queries)
    return qbuf, qoff, len(queries)


class search_no_errors:
    """search/SearchNoErrors.h"""

    @staticmethod
    def search(index, queries, out=None, want_stats=False):
        """returns (lb, len) arrays — cursor_t{lb, len} per query (len == 0: the reference reports nothing).
        `queries` = list of sequences or (qbuf, qoff); numpy arrays or DeviceBuffers.  `out` = (lb, len) DeviceBuffers
        to keep results in HBM."""
        qbuf, qoff, nq = _queries(queries)
        if out is None:
            lb, ln = np.empty(nq, dtype=np.uint64), np.empty(nq, dtype=np.uint64)
        else:
            lb, ln = out
        st = capi.Stats()
        capi.check(capi.lib().fmgpu_search_exact(index._h, capi.ptr(qbuf), capi.ptr(qoff), nq, capi.ptr(lb), capi.ptr(ln),
                                                 C.byref(st) if want_stats else None, None))
        return (lb, ln, st) if want_stats else (lb, ln)


    @staticmethod
    def depth(index, queries):
        """per query: symbols consumed until the cursor holds at most one row (length + 1: still several rows at the end)"""
        qbuf, qoff, nq = _queries(queries)
        out = np.empty(nq, dtype=np.uint32)
        capi.check(capi.lib().fmgpu_search_exact_depth(index._h, capi.ptr(qbuf), capi.ptr(qoff), nq, capi.ptr(out), None))
        return out

    @staticmethod
    def search_packed(index, queries, out=None, want_stats=False):
        """the same cursors as one word per query, lb << 32 | len (the form a rank's intervals are gathered in)"""
        qbuf, qoff, nq = _queries(queries)
        word = np.empty(nq, dtype=np.uint64) if out is None else out
        st = capi.Stats()
        capi.check(capi.lib().fmgpu_search_exact_packed(index._h, capi.ptr(qbuf), capi.ptr(qoff), nq, capi.ptr(word),
                                                        C.byref(st) if want_stats else None, None))
        return (word, st) if want_stats else word


def _run_hits(call, capacity):
    while True:
        out = np.zeros(max(capacity, 1), dtype=HIT_DTYPE)
        cnt = C.c_uint64()
        st = capi.Stats()
        rc = call(out, capacity, cnt, st)
        if rc == capi.FMGPU_ERR_CAPACITY:
            capacity = int(cnt.value)
            continue
        capi.check(rc)
        hits = np.ascontiguousarray(out[: cnt.value])
        # the reference invokes the delegate in ascending qidx, inside a query in DFS order: (qidx, seq) restores it (sorted on the device)
        capi.check(capi.lib().fmgpu_hits_sort(capi.ptr(hits), hits.size, None))
        return hits, st


class search_backtracking:
    """search/Backtracking.h"""

    @staticmethod
    def search(index, queries, max_errors, capacity=None, want_stats=False):
        qbuf, qoff, nq = _queries(queries)
        cap = capacity if capacity is not None else max(1024, 4 * nq)
        hits, st = _run_hits(lambda out, c, cnt, st: capi.lib().fmgpu_search_backtracking(
            index._h, capi.ptr(qbuf), capi.ptr(qoff), nq, max_errors, capi.ptr(out), c, C.byref(cnt), C.byref(st), None), cap)
        return (hits, st) if want_stats else hits


class search_ng26:
    """search/SearchNg26.h: Hamming distance (Edit = false, default here) or edit distance (edit=True, the reference's default)"""

    @staticmethod
    def search(index, queries, scheme, partition=None, n=UINT64_MAX, capacity=None, want_stats=False, edit=False):
        """scheme = (pi, l, u) arrays [searches][parts]; partition = explicit part lengths or None (uniform per query)"""
        qbuf, qoff, nq = _queries(queries)
        pi, l, u = (_u64(x) for x in scheme)
        sc = capi.Scheme()
        sc.n_searches, sc.n_parts = pi.shape
        sc.pi, sc.l, sc.u = (x.ctypes.data_as(capi.u64p) for x in (pi, l, u))
        part = _u64(partition) if partition is not None else None
        sc.partition = part.ctypes.data_as(capi.u64p) if part is not None else None
        sc.edit = 1 if edit else 0
        cap = capacity if capacity is not None else max(1024, 4 * nq)
        hits, st = _run_hits(lambda out, c, cnt, st: capi.lib().fmgpu_search_scheme(
            index._h, capi.ptr(qbuf), capi.ptr(qoff), nq, C.byref(sc), n, capi.ptr(out), c, C.byref(cnt), C.byref(st), None), cap)
        return (hits, st) if want_stats else hits


class Replicas:
    """one index file on several GPUs of this process (fmgpu_replicas_*; SURVEY 8b `fmgpu_set_devices`): a batch is cut into contiguous ranges, one per
    replica, searched concurrently; results land in host arrays in batch order.  devices=None: every visible device."""

    def __init__(self, handle):
        self._r = handle
        n = C.c_int32()
        capi.check(capi.lib().fmgpu_replicas_info(self._r, C.byref(n), None, 0, None))
        dev = (C.c_int32 * n.value)()
        capi.check(capi.lib().fmgpu_replicas_info(self._r, None, dev, n.value, None))
        self.devices = list(dev)
        pc = C.c_int32()
        capi.check(capi.lib().fmgpu_replicas_peer_copies(self._r, C.byref(pc)))
        self.peer_copies = pc.value                 # replicas made by a device-to-device copy of the first one (the file was read once)

    @classmethod
    def load(cls, path, devices=None):
        h = C.c_void_p()
        d = (C.c_int32 * len(devices))(*devices) if devices else None
        capi.check(capi.lib().fmgpu_replicas_load(os.fsencode(path), d, len(devices) if devices else 0, C.byref(h)))
        return cls(h)

    def search_exact(self, queries, want_stats=False):
        qbuf, qoff, nq = _queries(queries)
        lb, ln = np.empty(nq, dtype=np.uint64), np.empty(nq, dtype=np.uint64)
        st = capi.Stats()
        capi.check(capi.lib().fmgpu_replicas_search_exact(self._r, capi.ptr(qbuf), capi.ptr(qoff), nq, capi.ptr(lb), capi.ptr(ln), C.byref(st)))
        return (lb, ln, st) if want_stats else (lb, ln)

    def search_scheme(self, queries, scheme, partition=None, n=UINT64_MAX, capacity=None, want_stats=False, edit=False):
        qbuf, qoff, nq = _queries(queries)
        pi, l, u = (_u64(x) for x in scheme)
        sc = capi.Scheme()
        sc.n_searches, sc.n_parts = pi.shape
        sc.pi, sc.l, sc.u = (x.ctypes.data_as(capi.u64p) for x in (pi, l, u))
        part = _u64(partition) if partition is not None else None
        sc.partition = part.ctypes.data_as(capi.u64p) if part is not None else None
        sc.edit = 1 if edit else 0
        cap = capacity if capacity is not None else max(1024, 4 * nq)
        hits, st = _run_hits(lambda out, c, cnt, st: capi.lib().fmgpu_replicas_search_scheme(
            self._r, capi.ptr(qbuf), capi.ptr(qoff), nq, C.byref(sc), n, capi.ptr(out), c, C.byref(cnt), C.byref(st)), cap)
        return (hits, st) if want_stats else hits

    def search_ng21(self, queries, scheme, n=UINT64_MAX, capacity=None, want_stats=False):
        """search_ng21::search / search_n over an expanded scheme (pi, l, u arrays [searches][query length])"""
        qbuf, qoff, nq = _queries(queries)
        pi, l, u = (_u64(x) for x in scheme)
        sc = capi.ExpandedScheme()
        sc.n_searches, sc.length = (pi.shape if pi.ndim == 2 else (0, 0))
        sc.pi, sc.l, sc.u = (x.ctypes.data_as(capi.u64p) for x in (pi, l, u))
        cap = capacity if capacity is not None else max(1024, 4 * nq)
        hits, st = _run_hits(lambda out, c, cnt, st: capi.lib().fmgpu_replicas_search_ng21(
            self._r, capi.ptr(qbuf), capi.ptr(qoff), nq, C.byref(sc), n, capi.ptr(out), c, C.byref(cnt), C.byref(st)), cap)
        return (hits, st) if want_stats else hits

    def locate(self, rows):
        """(seq, pos, steps) of every row (FMIndex::locate), the rows sharded over the replicas"""
        rows = np.ascontiguousarray(rows, dtype=np.uint64)
        out = [np.empty(len(rows), dtype=np.uint64) for _ in range(3)]
        capi.check(capi.lib().fmgpu_replicas_locate(self._r, capi.ptr(rows), len(rows), capi.ptr(out[0]), capi.ptr(out[1]), capi.ptr(out[2]), None))
        return tuple(out)

    def close(self):
        if self._r:
            capi.check(capi.lib().fmgpu_replicas_destroy(self._r))
            self._r = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class search_ng21:
    """search/SearchNg21.h: edit-distance search over an EXPANDED scheme — search_scheme.expand(scheme, query length), one {pi, l, u} entry
    per query symbol, so all queries of a call have that length (the reference indexes query[pi[k]] without a check)"""

    @staticmethod
    def search(index, queries, scheme, capacity=None, want_stats=False, n=UINT64_MAX):
        """search_ng21::search (:205-217); scheme = (pi, l, u) arrays [searches][query length]"""
        qbuf, qoff, nq = _queries(queries)
        pi, l, u = (_u64(x) for x in scheme)
        sc = capi.ExpandedScheme()
        sc.n_searches, sc.length = (pi.shape if pi.ndim == 2 else (0, 0))
        sc.pi, sc.l, sc.u = (x.ctypes.data_as(capi.u64p) for x in (pi, l, u))
        cap = capacity if capacity is not None else max(1024, 4 * nq)
        hits, st = _run_hits(lambda out, c, cnt, st: capi.lib().fmgpu_search_ng21(
            index._h, capi.ptr(qbuf), capi.ptr(qoff), nq, C.byref(sc), n, capi.ptr(out), c, C.byref(cnt), C.byref(st), None), cap)
        return (hits, st) if want_stats else hits

    @staticmethod
    def search_n(index, queries, scheme, n, capacity=None, want_stats=False):
        """search_ng21::search_n (:220-240): at most n rows per query, the last cursor clipped"""
        return search_ng21.search(index, queries, scheme, capacity, want_stats, n)

    @staticmethod
    def search_best(index, queries, schemes, n=UINT64_MAX):
        """search_ng21::search_best (:242-264): per query the first scheme of the list that reports any row"""
        qbuf, qoff, nq = _queries(queries)
        if not isinstance(qoff, np.ndarray):
            qoff = qoff.to_array(np.uint64, nq + 1) if isinstance(qoff, DeviceBuffer) else np.asarray(qoff)
        if not isinstance(qbuf, np.ndarray):
            qbuf = qbuf.to_array(np.uint8, int(qoff[-1]))
        todo = np.arange(nq)
        parts = []
        for rank, sch in enumerate(schemes):
            if todo.size == 0:
                break
            qb, qo = flatten([qbuf[int(qoff[i]): int(qoff[i + 1])] for i in todo])
            hits = search_ng21.search(index, (qb, qo), sch, n=n).copy()
            rows = np.bincount(hits["qidx"].astype(np.int64), weights=hits["len"].astype(np.float64), minlength=todo.size)
            hits["qidx"] = todo.astype(np.uint64)[hits["qidx"].astype(np.int64)]
            parts.append((rank, hits))
            todo = todo[rows == 0]                                    # `if (ct > 0) break;` (:261)
        if not parts:
            return np.zeros(0, dtype=HIT_DTYPE)
        hits = np.concatenate([h for _, h in parts])
        order = np.concatenate([np.full(len(h), r) for r, h in parts])
        return hits[np.lexsort((hits["seq"], order, hits["qidx"]))]

    @staticmethod
    def search_best_n(index, queries, schemes, n):
        """search_ng21::search_best_n (:267-293)"""
        return search_ng21.search_best(index, queries, schemes, n)


def _auto_scheme_search(index, queries, errors, n, edit, compat_auto_scheme):
    """search_ng26::search<Edit>(index, queries, maxErrors, delegate, n) (search/SearchNg26.h:436-444): per query length the cached scheme
    h2(maxErrors + (length == 2 ? 1 : 2), 0, maxErrors) (CachedSearchScheme.h:16-36) with a uniform partition"""
    qbuf, qoff, nq = _queries(queries)
    if not isinstance(qoff, np.ndarray):                      # the facade splits the batch by length on the host
        qoff = qoff.to_array(np.uint64, nq + 1) if isinstance(qoff, DeviceBuffer) else np.asarray(qoff)
    if not isinstance(qbuf, np.ndarray):
        qbuf = qbuf.to_array(np.uint8, int(qoff[-1]))
    lens = np.diff(qoff.astype(np.int64))

    def scheme_for(short):
        sc = search_scheme.h2(errors + (1 if short else 2), 0, errors)
        return search_scheme.limitToHamming(sc) if (compat_auto_scheme and not edit) else sc

    parts = []
    for short in (False, True):
        sel = np.nonzero((lens == 2) == short)[0]
        if sel.size == 0:
            continue
        sub = [qbuf[int(qoff[i]): int(qoff[i + 1])] for i in sel] if sel.size != nq else None
        qb, qo = (qbuf, qoff) if sub is None else flatten(sub)
        hits = search_ng26.search(index, (qb, qo), scheme_for(short), None, n, edit=edit)
        if sub is not None:
            hits = hits.copy()
            hits["qidx"] = sel.astype(np.uint64)[hits["qidx"].astype(np.int64)]
        parts.append(hits)
    if not parts:
        return np.zeros(0, dtype=HIT_DTYPE)
    hits = np.concatenate(parts)
    return hits[np.lexsort((hits["seq"], hits["qidx"]))]


def search(index, queries, errors, n=UINT64_MAX, compat_auto_scheme=False, edit=False):
    """fmc::search<EditDistance> (search/search.h:26-35; edit=False: Hamming, edit=True: edit distance, the reference's default):
    errors == 0 -> search_no_errors, else search_ng26 with h2(errors+2, 0, errors) and a uniform partition.  For Hamming distance the
    reference's convenience overload additionally applies limitToHamming to the un-expanded scheme (search/CachedSearchScheme.h:26-30),
    which loses hits; compat_auto_scheme=True reproduces exactly that."""
    if errors == 0:
        lb, ln = search_no_errors.search(index, queries)
        keep = np.nonzero(ln)[0]
        hits = np.zeros(keep.size, dtype=HIT_DTYPE)
        hits["qidx"], hits["lb"], hits["len"] = keep, lb[keep], ln[keep]
        return hits
    return _auto_scheme_search(index, queries, errors, n, edit, compat_auto_scheme)


def search_n(index, queries, errors, n, edit=True, compat_auto_scheme=False):
    """fmc::search_n<EditDistance> (search/search.h:38-46): at most n rows per query, always through search_ng26 (also for errors == 0)"""
    return _auto_scheme_search(index, queries, errors, n, edit, compat_auto_scheme)


def search_best(index, queries, max_errors, n=UINT64_MAX, edit=True, schemes=None):
    """search_ng26::search_best (search/SearchNg26.h:447-487).
    schemes=None: the convenience overload (:476-487) — the whole batch is searched with 0, 1, ... max_errors - 1 errors (the loop ends
    BEFORE max_errors, as in the reference) and stops at the first error count for which ANY query reports a hit.
    schemes=[(scheme, partition), ...]: the explicit overload (:447-473) — per query the first scheme that reports anything wins."""
    if schemes is None:
        for k in range(int(max_errors)):
            hits = _auto_scheme_search(index, queries, k, n, edit, False)
            if len(hits):
                return hits
        return np.zeros(0, dtype=HIT_DTYPE)
    qbuf, qoff, nq = _queries(queries)
    if not isinstance(qoff, np.ndarray):
        qoff = qoff.to_array(np.uint64, nq + 1) if isinstance(qoff, DeviceBuffer) else np.asarray(qoff)
    if not isinstance(qbuf, np.ndarray):
        qbuf = qbuf.to_array(np.uint8, int(qoff[-1]))
    todo = np.arange(nq)
    parts = []
    for sch, part in schemes:
        if todo.size == 0:
            break
        qb, qo = flatten([qbuf[int(qoff[i]): int(qoff[i + 1])] for i in todo])
        hits = search_ng26.search(index, (qb, qo), sch, part, n, edit=edit).copy()
        found = np.unique(hits["qidx"].astype(np.int64))
        hits["qidx"] = todo.astype(np.uint64)[hits["qidx"].astype(np.int64)]
        parts.append(hits)
        todo = np.delete(todo, found)
    if not parts:
        return np.zeros(0, dtype=HIT_DTYPE)
    hits = np.concatenate(parts)
    return hits[np.lexsort((hits["seq"], hits["qidx"]))]


class LocateLinear:
    """locate.h:14-57: iterate a cursor's rows -> (seqId, pos, offset); batched over many cursors here"""

    def __init__(self, index, lb, length):
        self.index = index
        lb, length = _u64(lb).reshape(-1), _u64(length).reshape(-1)
        self.owner = np.repeat(np.arange(lb.size, dtype=np.uint64), length.astype(np.int64))
        starts = np.repeat(lb, length.astype(np.int64))
        first = np.repeat(np.cumsum(length) - length, length.astype(np.int64))
        self.rows = starts + (np.arange(self.owner.size, dtype=np.uint64) - first)

    def __call__(self):
        seq, pos, steps = self.index.locate(self.rows)
        return self.owner, seq, pos, steps
