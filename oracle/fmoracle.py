"""ctypes front-end of the CPU oracle (oracle/libfmoracle.so) and, where it has been built,
of the real-reference library (oracle/_ref/libfmref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg — never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FMORACLE_LIBRARY") or os.path.join(HERE, "libfmoracle.so")      # (FMORACLE_LIBRARY: the sanitizer build of `make -C oracle asan`)
REF_PATH = os.path.join(HERE, "_ref", "libfmref.so")

LAYOUTS = {
    "IB8": 0, "IB16": 1, "IB32": 2, "IB16A": 3, "IBP16": 4,
    "EPR8": 5, "EPR16": 6, "EPR32": 7,
    "EPRV2_8": 8, "EPRV2_16": 9, "EPRV2_32": 10, "WAVELET": 11,
    "EPRV3_8": 12, "EPRV3_16": 13, "EPRV3_32": 14, "EPRV4": 15, "EPRV5": 16, "IEPRV7": 17,
    "FBV_64_64K": 18, "FBV_512_64K": 19, "FBV_2048_64K": 20,
}
REF_UNBUILDABLE = ("FBV_64_64K", "FBV_512_64K", "FBV_2048_64K")    # FlattenedBitvectors2L.h includes ../utils.h (libsais / mmser): no live reference
HIER_LAYOUTS = ("EPRV3_8", "EPRV3_16", "EPRV3_32", "EPRV4", "EPRV5", "IEPRV7", "FBV_64_64K", "FBV_512_64K", "FBV_2048_64K")   # bit planes + counter levels in arrays of their own
UINT64_MAX = (1 << 64) - 1

u8p = C.POINTER(C.c_uint8)
u64p = C.POINTER(C.c_uint64)


def build(ref=True):
    """(re)build the oracle; the reference library only where /root/reference exists."""
    subprocess.run(["make", "-C", HERE, "-s", "libfmoracle.so"], check=True)
    if ref and os.path.isdir("/root/reference/src/fmindex-collection"):
        subprocess.run(["make", "-C", HERE, "-s", "ref"], check=True)


def _p8(a):
    return a.ctypes.data_as(u8p)


def _p64(a):
    return a.ctypes.data_as(u64p) if a is not None else None


class Hit(C.Structure):
    _fields_ = [("qidx", C.c_uint64), ("lb", C.c_uint64), ("lb_rev", C.c_uint64),
                ("len", C.c_uint64), ("errors", C.c_uint64)]


HIT_DTYPE = np.dtype([("qidx", "<u8"), ("lb", "<u8"), ("lb_rev", "<u8"), ("len", "<u8"), ("errors", "<u8")])


class DenseVector(C.Structure):
    _fields_ = [("data", u64p), ("nwords", C.c_uint64), ("bitCount", C.c_uint64), ("bits", C.c_uint8),
                ("largestValue", C.c_uint64), ("commonDivisor", C.c_uint64)]


class Sparse(C.Structure):
    _fields_ = [("n", C.c_uint64), ("l0", u64p), ("nl0", C.c_uint64),
                ("l1", C.POINTER(C.c_uint16)), ("nl1", C.c_uint64),
                ("bits", u64p), ("nbitwords", C.c_uint64),
                ("field", DenseVector * 2), ("nvalues", C.c_uint64)]


class IndexStruct(C.Structure):
    _fields_ = [("sigma", C.c_int), ("layout", C.c_int), ("bidirectional", C.c_int), ("n", C.c_uint64),
                ("bwt", C.c_void_p), ("bwt_rev", C.c_void_p), ("C", C.c_uint64 * 258), ("sa", C.POINTER(Sparse))]


class Cursor(C.Structure):
    _fields_ = [("lb", C.c_uint64), ("lb_rev", C.c_uint64), ("len", C.c_uint64)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build(ref=False)
        L = C.CDLL(LIB_PATH)
        L.ora_string_build.restype = C.c_void_p
        L.ora_string_build.argtypes = [C.c_int, C.c_int, u8p, C.c_uint64]
        L.ora_string_free.argtypes = [C.c_void_p]
        for f in ("ora_string_size", "ora_string_block_stride", "ora_string_bits_offset"):
            getattr(L, f).restype = C.c_uint64
            getattr(L, f).argtypes = [C.c_void_p]
        for f in ("ora_rank", "ora_prefix_rank"):
            getattr(L, f).restype = C.c_uint64
            getattr(L, f).argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
        L.ora_symbol.restype = C.c_uint64
        L.ora_symbol.argtypes = [C.c_void_p, C.c_uint64]
        L.ora_all_ranks_and_prefix_ranks.argtypes = [C.c_void_p, C.c_uint64, u64p, u64p]
        L.ora_string_raw.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), u64p]
        L.ora_sparse_build.restype = C.POINTER(Sparse)
        L.ora_sparse_build.argtypes = [C.c_uint64, u8p, u64p, u64p]
        L.ora_sparse_free.argtypes = [C.POINTER(Sparse)]
        L.ora_sparse_value.argtypes = [C.POINTER(Sparse), C.c_uint64, u64p, u64p]
        L.ora_sparse_rank.restype = C.c_uint64
        L.ora_sparse_rank.argtypes = [C.POINTER(Sparse), C.c_uint64]
        L.ora_dense_build.restype = C.POINTER(DenseVector)
        L.ora_dense_build.argtypes = [u64p, C.c_uint64, C.c_uint64, C.c_uint64]
        L.ora_dense_concat.restype = C.POINTER(DenseVector)
        L.ora_dense_concat.argtypes = [C.POINTER(DenseVector), C.POINTER(DenseVector)]
        L.ora_dense_free.argtypes = [C.POINTER(DenseVector)]
        L.ora_dense_size.restype = C.c_uint64
        L.ora_dense_size.argtypes = [C.POINTER(DenseVector)]
        L.ora_dense_access.restype = C.c_uint64
        L.ora_dense_access.argtypes = [C.POINTER(DenseVector), C.c_uint64]
        L.ora_suffix_array.argtypes = [u8p, C.c_uint64, u64p]
        L.ora_bwt_from_sa.argtypes = [u8p, C.c_uint64, u64p, u8p]
        L.ora_index_build.restype = C.POINTER(IndexStruct)
        L.ora_index_build.argtypes = [C.c_int, C.c_int, u8p, u64p, C.c_uint64, C.c_uint64, C.c_int]
        L.ora_index_from_bwt.restype = C.POINTER(IndexStruct)
        L.ora_index_from_bwt.argtypes = [C.c_int, C.c_int, u8p, u8p, C.c_uint64, u8p, u64p, u64p]
        L.ora_index_free.argtypes = [C.POINTER(IndexStruct)]
        L.ora_index_spread.argtypes = [C.POINTER(IndexStruct), C.c_int]
        L.ora_index_spread.restype = None
        L.ora_cursor_init.restype = Cursor
        L.ora_cursor_init.argtypes = [C.POINTER(IndexStruct)]
        for f in ("ora_extend_left", "ora_extend_right"):
            getattr(L, f).restype = Cursor
            getattr(L, f).argtypes = [C.POINTER(IndexStruct), Cursor, C.c_uint64]
        for f in ("ora_extend_left_all", "ora_extend_right_all"):
            getattr(L, f).argtypes = [C.POINTER(IndexStruct), Cursor, C.POINTER(Cursor)]
        L.ora_locate.argtypes = [C.POINTER(IndexStruct), C.c_uint64, u64p, u64p, u64p]
        L.ora_search_exact.argtypes = [C.POINTER(IndexStruct), u8p, u64p, C.c_uint64, u64p, u64p, u64p, C.c_int]
        L.ora_search_backtracking.restype = C.c_uint64
        L.ora_search_backtracking.argtypes = [C.POINTER(IndexStruct), u8p, u64p, C.c_uint64, C.c_uint64,
                                              C.c_void_p, C.c_uint64, u64p]
        L.ora_search_ng26_hamming.restype = C.c_uint64
        L.ora_search_ng26_hamming.argtypes = [C.POINTER(IndexStruct), u8p, u64p, C.c_uint64, C.c_int, C.c_int,
                                              u64p, u64p, u64p, u64p, C.c_uint64, C.c_void_p, C.c_uint64,
                                              u64p, u64p, C.c_int]
        L.ora_search_exact_batched.argtypes = [C.POINTER(IndexStruct), u8p, u64p, C.c_uint64, u64p, u64p, C.c_int, C.c_int]
        L.ora_search_ng21.restype = C.c_uint64
        L.ora_search_ng21.argtypes = [C.POINTER(IndexStruct), u8p, u64p, C.c_uint64, C.c_int, C.c_uint64, u64p, u64p, u64p,
                                      C.c_uint64, C.c_void_p, C.c_uint64, u64p, C.POINTER(C.c_uint64)]
        L.ora_search_ng26.restype = C.c_uint64
        L.ora_search_ng26.argtypes = [C.POINTER(IndexStruct), C.c_int, u8p, u64p, C.c_uint64, C.c_int, C.c_int,
                                      u64p, u64p, u64p, u64p, C.c_uint64, C.c_void_p, C.c_uint64, u64p, u64p, C.c_int]
        for f in ("ora_scheme_h2", "ora_scheme_backtracking"):
            getattr(L, f).argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, u64p, u64p, u64p]
        for f in ("ora_scheme_pigeon_opt", "ora_scheme_pigeon_trivial"):
            getattr(L, f).argtypes = [C.c_uint64, C.c_uint64, u64p, u64p, u64p]
        L.ora_uniform_partition.argtypes = [C.c_uint64, C.c_uint64, u64p]
        L.ora_scheme_expand.argtypes = [C.c_int, C.c_uint64, u64p, u64p, u64p, C.c_uint64, u64p, u64p, u64p]
        L.ora_scheme_limit_to_hamming.argtypes = [C.c_int, C.c_uint64, u64p, u64p]
        L.ora_scheme_is_valid.argtypes = [C.c_int, C.c_uint64, u64p, u64p, u64p]
        L.ora_scheme_is_complete.argtypes = [C.c_int, C.c_uint64, u64p, u64p, u64p, C.c_uint64, C.c_uint64]
        L.ora_scheme_node_count_hamming.restype = C.c_double
        L.ora_scheme_node_count_hamming.argtypes = [C.c_int, C.c_uint64, u64p, u64p, C.c_uint64]
        _lib = L
    return _lib


def as_u8(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.uint8))


def as_u64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.uint64))


def flatten_queries(queries):
    """list of sequences -> (qbuf u8, qoff u64[nq+1])"""
    lens = np.array([len(q) for q in queries], dtype=np.uint64)
    qoff = np.zeros(len(queries) + 1, dtype=np.uint64)
    np.cumsum(lens, out=qoff[1:])
    qbuf = np.zeros(max(int(qoff[-1]), 1), dtype=np.uint8)
    if len(queries) and int(qoff[-1]):
        qbuf[:int(qoff[-1])] = np.concatenate([as_u8(q) for q in queries if len(q)])
    return qbuf, qoff


class _StringBase:
    """common raw-array helpers of OraString / RefString"""

    def raw(self, part):
        ptr = C.c_void_p()
        nbytes = C.c_uint64()
        if self._raw(self.h, part, C.byref(ptr), C.byref(nbytes)) != 0:
            return None
        if nbytes.value == 0:
            return np.zeros(0, dtype=np.uint8)
        buf = (C.c_uint8 * nbytes.value).from_address(ptr.value)
        return np.frombuffer(buf, dtype=np.uint8).copy()

    def level_fields(self):
        """EPRV3/4/5/7: the raw arrays [bits, superBlocks, level0 / blocks_, level1, level2] (missing ones empty)"""
        out = []
        for part in range(5):
            r = self.raw(part)
            out.append(r if r is not None else np.zeros(0, dtype=np.uint8))
        return out

    def block_fields(self):
        """blocked layouts: (counts [nblocks, sigma], words [nblocks, K] u64, superBlocks [k, sigma] u64)"""
        if self.layout in HIER_LAYOUTS:
            return tuple(self.level_fields())
        bt = {"IB8": 1, "IB16": 2, "IB32": 4, "IB16A": 2, "IBP16": 2, "EPR8": 1, "EPR16": 2, "EPR32": 4,
              "EPRV2_8": 1, "EPRV2_16": 2, "EPRV2_32": 4}[self.layout]
        stride = self.block_stride()
        raw = self.raw(0)
        nb = len(raw) // stride
        rows = raw.reshape(nb, stride)
        cnt = rows[:, : self.sigma * bt].copy().view({1: "<u1", 2: "<u2", 4: "<u4"}[bt]).astype(np.uint64)
        off = (self.sigma * bt + 7) // 8 * 8
        bitct = max(1, (self.sigma - 1).bit_length())
        K = {"IB": self.sigma, "EP": 1}[self.layout[:2]]
        if self.layout.startswith("EPRV2"):
            K = bitct
        words = rows[:, off: off + 8 * K].copy().view("<u8")
        sup = self.raw(1).view("<u8").reshape(-1, self.sigma)
        return cnt, words, sup


class OraString(_StringBase):
    def __init__(self, layout, sigma, symbols):
        self.layout, self.sigma = layout, sigma
        self.symbols = as_u8(symbols)
        self.h = lib().ora_string_build(LAYOUTS[layout], sigma, _p8(self.symbols), len(self.symbols))
        if not self.h:
            raise ValueError("ora_string_build failed")
        self._raw = lib().ora_string_raw

    def __del__(self):
        if getattr(self, "h", None):
            lib().ora_string_free(self.h)
            self.h = None

    def size(self):
        return lib().ora_string_size(self.h)

    def rank(self, i, c):
        return lib().ora_rank(self.h, i, c)

    def prefix_rank(self, i, c):
        return lib().ora_prefix_rank(self.h, i, c)

    def symbol(self, i):
        return lib().ora_symbol(self.h, i)

    def block_stride(self):
        return lib().ora_string_block_stride(self.h)

    def all_ranks_and_prefix_ranks(self, i):
        rs = np.zeros(self.sigma, dtype=np.uint64)
        prs = np.zeros(self.sigma, dtype=np.uint64)
        lib().ora_all_ranks_and_prefix_ranks(self.h, i, _p64(rs), _p64(prs))
        return rs, prs

    def rank_table(self):
        n = self.size()
        r = np.zeros((n + 1, self.sigma), dtype=np.uint64)
        p = np.zeros((n + 1, self.sigma), dtype=np.uint64)
        L = lib()
        for i in range(n + 1):
            for c in range(self.sigma):
                r[i, c] = L.ora_rank(self.h, i, c)
                p[i, c] = L.ora_prefix_rank(self.h, i, c)
        return r, p


class OraDenseVector:
    """DenseVector (DenseVector.h:26-203): DenseVector{values}, or DenseVector(largest, divisor) followed by push_back of the values"""

    def __init__(self, values=None, largest=0, divisor=0, _ptr=None):
        if _ptr is None:
            v = as_u64(values)
            _ptr = lib().ora_dense_build(_p64(v), len(v), largest, divisor)
        self.p = _ptr

    def __del__(self):
        if getattr(self, "p", None):
            lib().ora_dense_free(self.p)
            self.p = None

    @classmethod
    def concat(cls, a, b):
        return cls(_ptr=lib().ora_dense_concat(a.p, b.p))

    def __len__(self):
        return int(lib().ora_dense_size(self.p))

    def __getitem__(self, i):
        return int(lib().ora_dense_access(self.p, i))

    @property
    def common_divisor(self):
        return int(self.p.contents.commonDivisor)

    @property
    def bits(self):
        return int(self.p.contents.bits)


class OraSparse:
    """SparseArray<tuple, Bitvector2L<512, 65536>> (suffixarray/SparseArray.h:31-76) on its own: has[i] marks the rows that carry (seq[i], pos[i])"""

    def __init__(self, has, seq=None, pos=None):
        self.has = as_u8(has)
        n = len(self.has)
        self.seq = as_u64(seq if seq is not None else np.zeros(n))
        self.pos = as_u64(pos if pos is not None else np.zeros(n))
        self.p = lib().ora_sparse_build(n, _p8(self.has), _p64(self.seq), _p64(self.pos))

    def __del__(self):
        if getattr(self, "p", None):
            lib().ora_sparse_free(self.p)
            self.p = None

    def rank(self, i):
        return int(lib().ora_sparse_rank(self.p, i))

    def value(self, i):
        a, b = C.c_uint64(), C.c_uint64()
        return (int(a.value), int(b.value)) if lib().ora_sparse_value(self.p, i, C.byref(a), C.byref(b)) else None


class OraIndex:
    """FMIndex / BiFMIndex restatement"""

    def __init__(self, ptr, keep=()):
        self.p = ptr
        self._keep = keep
        if not ptr:
            raise ValueError("index construction failed")

    @classmethod
    def build(cls, layout, sigma, sequences, sampling_rate=1, bidirectional=False):
        qbuf, qoff = flatten_queries(sequences)
        p = lib().ora_index_build(LAYOUTS[layout], sigma, _p8(qbuf), _p64(qoff), len(sequences), sampling_rate,
                                  1 if bidirectional else 0)
        return cls(p)

    @classmethod
    def from_bwt(cls, layout, sigma, bwt, bwt_rev, has, seq, pos):
        bwt = as_u8(bwt)
        bwt_rev = as_u8(bwt_rev) if bwt_rev is not None else None
        has_ = as_u8(has) if has is not None else None
        seq_ = as_u64(seq) if seq is not None else None
        pos_ = as_u64(pos) if pos is not None else None
        p = lib().ora_index_from_bwt(LAYOUTS[layout], sigma, _p8(bwt), _p8(bwt_rev) if bwt_rev is not None else None,
                                     len(bwt), _p8(has_) if has_ is not None else None, _p64(seq_), _p64(pos_))
        return cls(p, keep=(bwt, bwt_rev, has_, seq_, pos_))

    def __del__(self):
        if getattr(self, "p", None):
            lib().ora_index_free(self.p)
            self.p = None

    def spread(self, nthreads):
        """re-home the occurrence tables over the NUMA nodes of `nthreads` OpenMP threads (bench.py's all-core baseline)"""
        lib().ora_index_spread(self.p, int(nthreads))
        return self

    @property
    def n(self):
        return self.p.contents.n

    @property
    def sigma(self):
        return self.p.contents.sigma

    @property
    def bidirectional(self):
        return bool(self.p.contents.bidirectional)

    @property
    def C(self):
        return np.array(self.p.contents.C[: self.sigma + 1], dtype=np.uint64)

    def bwt_string(self, rev=False):
        s = _BorrowedString(self.p.contents.bwt_rev if rev else self.p.contents.bwt, self)
        return s

    def sparse(self):
        """the SparseArray parts as numpy arrays (copies)"""
        sp = self.p.contents.sa.contents
        out = {"n": sp.n, "nvalues": sp.nvalues,
               "l0": np.ctypeslib.as_array(sp.l0, (sp.nl0,)).copy(),
               "l1": np.ctypeslib.as_array(sp.l1, (sp.nl1,)).copy(),
               "bits": np.ctypeslib.as_array(sp.bits, (sp.nbitwords,)).copy(), "fields": []}
        for f in range(2):
            d = sp.field[f]
            out["fields"].append({"data": np.ctypeslib.as_array(d.data, (max(d.nwords, 1),)).copy()[: d.nwords],
                                  "bitCount": d.bitCount, "bits": d.bits,
                                  "largestValue": d.largestValue, "commonDivisor": d.commonDivisor})
        return out

    def cursor(self):
        return lib().ora_cursor_init(self.p)

    def extend_left(self, cur, c):
        return lib().ora_extend_left(self.p, cur, c)

    def extend_right(self, cur, c):
        return lib().ora_extend_right(self.p, cur, c)

    def extend_left_all(self, cur):
        out = (Cursor * self.sigma)()
        lib().ora_extend_left_all(self.p, cur, out)
        return list(out)

    def extend_right_all(self, cur):
        out = (Cursor * self.sigma)()
        lib().ora_extend_right_all(self.p, cur, out)
        return list(out)

    def locate(self, row):
        s, p, st = C.c_uint64(), C.c_uint64(), C.c_uint64()
        lib().ora_locate(self.p, row, C.byref(s), C.byref(p), C.byref(st))
        return s.value, p.value, st.value

    def single_locate_step(self, row):
        s, p = C.c_uint64(), C.c_uint64()
        if lib().ora_sparse_value(self.p.contents.sa, row, C.byref(s), C.byref(p)):
            return s.value, p.value
        return None

    def search_exact(self, qbuf, qoff, nthreads=1, want_steps=False):
        nq = len(qoff) - 1
        lb = np.zeros(nq, dtype=np.uint64)
        ln = np.zeros(nq, dtype=np.uint64)
        st = np.zeros(nq, dtype=np.uint64) if want_steps else None
        lib().ora_search_exact(self.p, _p8(qbuf), _p64(qoff), nq, _p64(lb), _p64(ln), _p64(st), nthreads)
        return (lb, ln, st) if want_steps else (lb, ln)

    def search_exact_batched(self, qbuf, qoff, batch=32, nthreads=1):
        nq = len(qoff) - 1
        lb = np.zeros(nq, dtype=np.uint64)
        ln = np.zeros(nq, dtype=np.uint64)
        lib().ora_search_exact_batched(self.p, _p8(qbuf), _p64(qoff), nq, _p64(lb), _p64(ln), batch, nthreads)
        return lb, ln

    def search_backtracking(self, qbuf, qoff, k, cap=1 << 20):
        nq = len(qoff) - 1
        out = np.zeros(cap, dtype=HIT_DTYPE)
        nodes = C.c_uint64()
        n = lib().ora_search_backtracking(self.p, _p8(qbuf), _p64(qoff), nq, k, out.ctypes.data, cap, C.byref(nodes))
        if n > cap:
            return self.search_backtracking(qbuf, qoff, k, cap=int(n))
        return out[:n], nodes.value

    def search_ng26(self, qbuf, qoff, scheme, partition=None, max_hits=UINT64_MAX, cap=1 << 20, nthreads=1, edit=None, records=False):
        """edit=None: the Hamming reduction; edit=False / True: the full state machine with Edit = false / true.
        nthreads > 1 counts only (the timed baseline) unless records=True: then a second threaded pass also fills the hit records"""
        if edit is not None:
            return self._search_ng26_full(qbuf, qoff, scheme, partition, max_hits, cap, nthreads, bool(edit))
        pi, l, u = scheme
        nsearch, nparts = pi.shape
        nq = len(qoff) - 1
        out = np.zeros(cap, dtype=HIT_DTYPE)
        qcount = np.zeros(nq, dtype=np.uint64)
        nodes = C.c_uint64()
        part = as_u64(partition) if partition is not None else None
        want = nthreads == 1 or records
        n = lib().ora_search_ng26_hamming(self.p, _p8(qbuf), _p64(qoff), nq, nsearch, nparts,
                                          _p64(as_u64(pi)), _p64(as_u64(l)), _p64(as_u64(u)), _p64(part),
                                          max_hits, out.ctypes.data if want else None,
                                          cap if want else 0, _p64(qcount), C.byref(nodes), nthreads)
        if want and n > cap:
            return self.search_ng26(qbuf, qoff, scheme, partition, max_hits, cap=int(n), nthreads=nthreads, records=records)
        return out[: n if want else 0], qcount, nodes.value


    def _search_ng26_full(self, qbuf, qoff, scheme, partition, max_hits, cap, nthreads, edit):
        pi, l, u = scheme
        nsearch, nparts = pi.shape
        nq = len(qoff) - 1
        out = np.zeros(cap, dtype=HIT_DTYPE)
        qcount = np.zeros(nq, dtype=np.uint64)
        nodes = C.c_uint64()
        part = as_u64(partition) if partition is not None else None
        n = lib().ora_search_ng26(self.p, 1 if edit else 0, _p8(qbuf), _p64(qoff), nq, nsearch, nparts,
                                  _p64(as_u64(pi)), _p64(as_u64(l)), _p64(as_u64(u)), _p64(part),
                                  max_hits, out.ctypes.data if nthreads == 1 else None,
                                  cap if nthreads == 1 else 0, _p64(qcount), C.byref(nodes), nthreads)
        if nthreads == 1 and n > cap:
            return self._search_ng26_full(qbuf, qoff, scheme, partition, max_hits, int(n), 1, edit)
        return out[: n if nthreads == 1 else 0], qcount, nodes.value


    def search_ng21(self, qbuf, qoff, expanded, max_hits=UINT64_MAX, cap=1 << 20):
        """search_ng21::search / search_n over an expanded scheme (pi, l, u of shape [searches, query length])"""
        pi, l, u = expanded
        nsearch, length = pi.shape if pi.ndim == 2 else (0, 0)
        nq = len(qoff) - 1
        out = np.zeros(cap, dtype=HIT_DTYPE)
        qcount = np.zeros(nq, dtype=np.uint64)
        nodes = C.c_uint64()
        n = lib().ora_search_ng21(self.p, _p8(qbuf), _p64(qoff), nq, nsearch, length, _p64(as_u64(pi)), _p64(as_u64(l)), _p64(as_u64(u)),
                                  max_hits, out.ctypes.data, cap, _p64(qcount), C.byref(nodes))
        if n > cap:
            return self.search_ng21(qbuf, qoff, expanded, max_hits, cap=int(n))
        return out[:n], qcount, nodes.value

    def search_ng21_best(self, qbuf, qoff, expanded_list, max_hits=UINT64_MAX):
        """search_best / search_best_n, SearchNg21.h:242-293: per query the first scheme of the list with any hit"""
        nq = len(qoff) - 1
        parts, nodes = [], 0
        todo = np.arange(nq)
        for ex in expanded_list:
            if len(todo) == 0:
                break
            sub_off = np.zeros(len(todo) + 1, dtype=np.uint64)
            lens = (qoff[1:] - qoff[:-1])[todo]
            sub_off[1:] = np.cumsum(lens)
            sub_buf = np.concatenate([qbuf[int(qoff[q]): int(qoff[q + 1])] for q in todo]) if len(todo) else np.zeros(0, np.uint8)
            hits, qcount, nd = self.search_ng21(np.ascontiguousarray(sub_buf, dtype=np.uint8), sub_off, ex, max_hits)
            nodes += nd
            hits = hits.copy()
            hits["qidx"] = todo[hits["qidx"].astype(np.int64)]
            parts.append(hits)
            todo = todo[qcount == 0]
        allh = np.concatenate(parts) if parts else np.zeros(0, dtype=HIT_DTYPE)
        return allh[np.argsort(allh["qidx"], kind="stable")], nodes


class _BorrowedString(OraString):
    def __init__(self, h, owner):
        self.h = h
        self._owner = owner
        self.sigma = owner.sigma
        self.layout = {v: k for k, v in LAYOUTS.items()}[owner.p.contents.layout]
        self._raw = lib().ora_string_raw

    def __del__(self):
        self.h = None


# ------------------------------------------------------------------------------ schemes
def _scheme_call(fn, nsearch_cap, parts, *args):
    pi = np.zeros((nsearch_cap, parts), dtype=np.uint64)
    l = np.zeros_like(pi)
    u = np.zeros_like(pi)
    n = fn(*args, _p64(pi), _p64(l), _p64(u))
    return pi[:n].copy(), l[:n].copy(), u[:n].copy()


def scheme_h2(N, minK, K):
    return _scheme_call(lib().ora_scheme_h2, K + 1, N, N, minK, K)


def scheme_pigeon_opt(minK, K):
    return _scheme_call(lib().ora_scheme_pigeon_opt, K + 1, K + 1, minK, K)


def scheme_pigeon_trivial(minK, K):
    return _scheme_call(lib().ora_scheme_pigeon_trivial, K + 1, K + 1, minK, K)


def scheme_backtracking(N, minK, K):
    return _scheme_call(lib().ora_scheme_backtracking, 1, N, N, minK, K)


def uniform_partition(parts, total):
    out = np.zeros(parts, dtype=np.uint64)
    lib().ora_uniform_partition(parts, total, _p64(out))
    return out


def scheme_expand(scheme, new_len):
    pi, l, u = (as_u64(x) for x in scheme)
    ns, parts = pi.shape
    opi = np.zeros((ns, new_len), dtype=np.uint64)
    ol = np.zeros_like(opi)
    ou = np.zeros_like(opi)
    k = lib().ora_scheme_expand(ns, parts, _p64(pi), _p64(l), _p64(u), new_len, _p64(opi), _p64(ol), _p64(ou))
    return opi[:k].copy(), ol[:k].copy(), ou[:k].copy()


def scheme_limit_to_hamming(scheme):
    pi, l, u = (as_u64(x).copy() for x in scheme)
    lib().ora_scheme_limit_to_hamming(pi.shape[0], pi.shape[1], _p64(l), _p64(u))
    return pi, l, u


def scheme_is_valid(scheme):
    pi, l, u = (as_u64(x) for x in scheme)
    return bool(lib().ora_scheme_is_valid(pi.shape[0], pi.shape[1], _p64(pi), _p64(l), _p64(u)))


def scheme_is_complete(scheme, minK, maxK):
    pi, l, u = (as_u64(x) for x in scheme)
    return bool(lib().ora_scheme_is_complete(pi.shape[0], pi.shape[1], _p64(pi), _p64(l), _p64(u), minK, maxK))


def scheme_node_count_hamming(scheme, sigma):
    pi, l, u = (as_u64(x) for x in scheme)
    return lib().ora_scheme_node_count_hamming(pi.shape[0], pi.shape[1], _p64(l), _p64(u), sigma)


# ------------------------------------------------------------------------------ the real reference (where built)
_ref = None


def ref_available():
    return os.path.exists(REF_PATH)


def ref():
    global _ref
    if _ref is None:
        R = C.CDLL(REF_PATH)
        R.fmref_string_create.restype = C.c_void_p
        R.fmref_string_create.argtypes = [C.c_int, C.c_int, u8p, C.c_uint64]
        R.fmref_string_destroy.argtypes = [C.c_void_p]
        R.fmref_string_size.restype = C.c_uint64
        R.fmref_string_size.argtypes = [C.c_void_p]
        for f in ("fmref_string_rank", "fmref_string_prefix_rank"):
            getattr(R, f).restype = C.c_uint64
            getattr(R, f).argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
        R.fmref_string_symbol.restype = C.c_uint64
        R.fmref_string_symbol.argtypes = [C.c_void_p, C.c_uint64]
        R.fmref_string_all_ranks_and_prefix_ranks.argtypes = [C.c_void_p, C.c_uint64, u64p, u64p]
        R.fmref_string_rank_table.argtypes = [C.c_void_p, C.c_int, u64p, u64p]
        R.fmref_string_raw.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), u64p]
        R.fmref_string_block_stride.restype = C.c_uint64
        R.fmref_string_block_stride.argtypes = [C.c_void_p]
        sig = [u64p, u64p, u64p, C.c_uint64, u64p]
        R.fmref_scheme_h2.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64] + sig
        R.fmref_scheme_backtracking.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64] + sig
        R.fmref_scheme_pigeon_opt.argtypes = [C.c_uint64, C.c_uint64] + sig
        R.fmref_scheme_pigeon_trivial.argtypes = [C.c_uint64, C.c_uint64] + sig
        R.fmref_scheme_expand.argtypes = [C.c_int, C.c_uint64, u64p, u64p, u64p, C.c_uint64] + sig
        R.fmref_scheme_limit_to_hamming.argtypes = [C.c_int, C.c_uint64, u64p, u64p, u64p] + sig
        R.fmref_scheme_is_valid.argtypes = [C.c_int, C.c_uint64, u64p, u64p, u64p]
        R.fmref_scheme_is_complete.argtypes = [C.c_int, C.c_uint64, u64p, u64p, u64p, C.c_uint64, C.c_uint64]
        R.fmref_scheme_node_count_hamming.restype = C.c_double
        R.fmref_scheme_node_count_hamming.argtypes = [C.c_int, C.c_uint64, u64p, u64p, u64p, C.c_uint64]
        R.fmref_uniform_partition.argtypes = [C.c_uint64, C.c_uint64, u64p]
        if hasattr(R, "fmref_scheme_expand_by_wnc"):
            R.fmref_scheme_expand_by_wnc.argtypes = [C.c_int, C.c_uint64, u64p, u64p, u64p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int] + sig
            R.fmref_scheme_weighted_node_count.restype = C.c_double
            R.fmref_scheme_weighted_node_count.argtypes = [C.c_int, C.c_uint64, u64p, u64p, u64p, C.c_uint64, C.c_uint64, C.c_int]
        _ref = R
    return _ref


class RefString(_StringBase):
    """a fmc::string::X<Sigma> of the real reference"""

    def __init__(self, layout, sigma, symbols):
        self.layout, self.sigma = layout, sigma
        self.symbols = as_u8(symbols)
        self.h = ref().fmref_string_create(LAYOUTS[layout], sigma, _p8(self.symbols), len(self.symbols))
        if not self.h:
            raise ValueError("unsupported (layout, sigma) in ref_driver.cpp")
        self._raw = ref().fmref_string_raw

    def __del__(self):
        if getattr(self, "h", None):
            ref().fmref_string_destroy(self.h)
            self.h = None

    def size(self):
        return ref().fmref_string_size(self.h)

    def rank(self, i, c):
        return ref().fmref_string_rank(self.h, i, c)

    def prefix_rank(self, i, c):
        return ref().fmref_string_prefix_rank(self.h, i, c)

    def symbol(self, i):
        return ref().fmref_string_symbol(self.h, i)

    def block_stride(self):
        return ref().fmref_string_block_stride(self.h)

    def rank_table(self):
        n = self.size()
        r = np.zeros((n + 1, self.sigma), dtype=np.uint64)
        p = np.zeros((n + 1, self.sigma), dtype=np.uint64)
        ref().fmref_string_rank_table(self.h, self.sigma, _p64(r), _p64(p))
        return r, p


def _ref_scheme(fn, *args, cap=4096):
    pi = np.zeros(cap, dtype=np.uint64)
    l = np.zeros(cap, dtype=np.uint64)
    u = np.zeros(cap, dtype=np.uint64)
    parts = C.c_uint64()
    n = fn(*args, _p64(pi), _p64(l), _p64(u), cap, C.byref(parts))
    assert n >= 0
    p = parts.value
    return (pi[: n * p].reshape(n, p).copy(), l[: n * p].reshape(n, p).copy(), u[: n * p].reshape(n, p).copy())


def ref_scheme_h2(N, minK, K):
    return _ref_scheme(ref().fmref_scheme_h2, N, minK, K)


def ref_scheme_pigeon_opt(minK, K):
    return _ref_scheme(ref().fmref_scheme_pigeon_opt, minK, K)


def ref_scheme_pigeon_trivial(minK, K):
    return _ref_scheme(ref().fmref_scheme_pigeon_trivial, minK, K)


def ref_scheme_backtracking(N, minK, K):
    return _ref_scheme(ref().fmref_scheme_backtracking, N, minK, K)


def ref_scheme_expand(scheme, new_len):
    pi, l, u = (as_u64(x) for x in scheme)
    return _ref_scheme(ref().fmref_scheme_expand, pi.shape[0], pi.shape[1], _p64(pi), _p64(l), _p64(u), new_len, cap=1 << 16)


def ref_scheme_limit_to_hamming(scheme):
    pi, l, u = (as_u64(x) for x in scheme)
    return _ref_scheme(ref().fmref_scheme_limit_to_hamming, pi.shape[0], pi.shape[1], _p64(pi), _p64(l), _p64(u), cap=1 << 16)


def ref_scheme_is_valid(scheme):
    pi, l, u = (as_u64(x) for x in scheme)
    return bool(ref().fmref_scheme_is_valid(pi.shape[0], pi.shape[1], _p64(pi), _p64(l), _p64(u)))


def ref_scheme_is_complete(scheme, minK, maxK):
    pi, l, u = (as_u64(x) for x in scheme)
    return bool(ref().fmref_scheme_is_complete(pi.shape[0], pi.shape[1], _p64(pi), _p64(l), _p64(u), minK, maxK))


def ref_scheme_node_count_hamming(scheme, sigma):
    pi, l, u = (as_u64(x) for x in scheme)
    return ref().fmref_scheme_node_count_hamming(pi.shape[0], pi.shape[1], _p64(pi), _p64(l), _p64(u), sigma)


def ref_scheme_expand_by_wnc(scheme, new_len, sigma, N, edit):
    pi, l, u = (as_u64(x) for x in scheme)
    return _ref_scheme(ref().fmref_scheme_expand_by_wnc, pi.shape[0], pi.shape[1], _p64(pi), _p64(l), _p64(u), new_len, sigma, N, 1 if edit else 0, cap=1 << 16)


def ref_scheme_weighted_node_count(scheme, sigma, N, edit):
    pi, l, u = (as_u64(x) for x in scheme)
    return ref().fmref_scheme_weighted_node_count(pi.shape[0], pi.shape[1], _p64(pi), _p64(l), _p64(u), sigma, N, 1 if edit else 0)


def ref_uniform_partition(parts, total):
    out = np.zeros(parts, dtype=np.uint64)
    ref().fmref_uniform_partition(parts, total, _p64(out))
    return out
