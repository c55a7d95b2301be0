"""ctypes binding of libfmgpu.so (include/fmgpu.h).  No fallback: if the HIP library is missing, importing fails loudly."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FMGPU_LIBRARY") or os.path.join(HERE, "libfmgpu.so")   # (FMGPU_LIBRARY: another build of the same ABI, e.g. a tuning variant)

u8p = C.POINTER(C.c_uint8)
u16p = C.POINTER(C.c_uint16)
u64p = C.POINTER(C.c_uint64)

LAYOUTS = {
    "IB8": 0, "IB16": 1, "IB32": 2, "IB16A": 3, "IBP16": 4,
    "EPR8": 5, "EPR16": 6, "EPR32": 7,
    "EPRV2_8": 8, "EPRV2_16": 9, "EPRV2_32": 10, "WAVELET": 11,
    "EPRV3_8": 12, "EPRV3_16": 13, "EPRV3_32": 14, "EPRV4": 15, "EPRV5": 16, "IEPRV7": 17,
    "FBV_64_64K": 18, "FBV_512_64K": 19, "FBV_2048_64K": 20,
}
LAYOUT_NAMES = {v: k for k, v in LAYOUTS.items()}
UINT64_MAX = (1 << 64) - 1

FMGPU_OK = 0
FMGPU_ERR_INVALID = -1
FMGPU_ERR_UNSUPPORTED = -2
FMGPU_ERR_HIP = -3
FMGPU_ERR_NO_DEVICE = -4
FMGPU_ERR_CAPACITY = -5
FMGPU_ERR_NOMEM = -6


class FmgpuError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"fmgpu error {code}: {msg}")
        self.code = code


class WaveletNode(C.Structure):
    _fields_ = [("superblocks", u64p), ("n_superblocks", C.c_uint64), ("blocks", u8p), ("n_blocks", C.c_uint64),
                ("bits", u64p), ("n_bits", C.c_uint64), ("total_length", C.c_uint64)]


class StringDesc(C.Structure):
    _fields_ = [("layout", C.c_int32), ("sigma", C.c_int32), ("n", C.c_uint64),
                ("blocks", C.c_void_p), ("blocks_bytes", C.c_uint64),
                ("super_blocks", u64p), ("n_super_blocks", C.c_uint64),
                ("nodes", C.POINTER(WaveletNode)), ("n_nodes", C.c_uint64),
                ("levels", C.c_void_p * 3), ("level_bytes", C.c_uint64 * 3)]


class DenseVectorDesc(C.Structure):
    _fields_ = [("data", u64p), ("n_words", C.c_uint64), ("bit_count", C.c_uint64), ("bits", C.c_uint32),
                ("largest_value", C.c_uint64), ("common_divisor", C.c_uint64)]


class SparseArrayDesc(C.Structure):
    _fields_ = [("n", C.c_uint64), ("l0", u64p), ("n_l0", C.c_uint64), ("l1", u16p), ("n_l1", C.c_uint64),
                ("bits", u64p), ("n_bit_words", C.c_uint64), ("field", DenseVectorDesc * 2)]


class IndexDesc(C.Structure):
    _fields_ = [("bwt", StringDesc), ("bwt_rev", C.POINTER(StringDesc)), ("C", u64p),
                ("annotated_array", C.POINTER(SparseArrayDesc))]


class Hit(C.Structure):
    _fields_ = [("qidx", C.c_uint64), ("lb", C.c_uint64), ("lb_rev", C.c_uint64), ("len", C.c_uint64),
                ("errors", C.c_uint32), ("seq", C.c_uint32)]


HIT_DTYPE = np.dtype([("qidx", "<u8"), ("lb", "<u8"), ("lb_rev", "<u8"), ("len", "<u8"), ("errors", "<u4"), ("seq", "<u4")])
assert HIT_DTYPE.itemsize == C.sizeof(Hit) == 40


class Scheme(C.Structure):
    _fields_ = [("n_searches", C.c_int32), ("n_parts", C.c_int32), ("pi", u64p), ("l", u64p), ("u", u64p),
                ("partition", u64p), ("edit", C.c_int32), ("reserved", C.c_int32)]


class ExpandedScheme(C.Structure):
    _fields_ = [("n_searches", C.c_int32), ("reserved", C.c_int32), ("length", C.c_uint64), ("pi", u64p), ("l", u64p), ("u", u64p)]


class Stats(C.Structure):
    _fields_ = [("lf_steps", C.c_uint64), ("hits", C.c_uint64), ("kernel_ms", C.c_float), ("prepass_ms", C.c_float),
                ("table_bytes", C.c_uint64), ("table_accesses", C.c_uint64), ("table_steps", C.c_uint64)]


# every symbol include/fmgpu.h declares (tests check that the library exports all of them)
EXPORTS = [
    "fmgpu_abi_version", "fmgpu_last_error", "fmgpu_device_count", "fmgpu_set_device",
    "fmgpu_index_create", "fmgpu_index_destroy", "fmgpu_index_info", "fmgpu_string_query",
    "fmgpu_search_exact", "fmgpu_search_exact_packed", "fmgpu_search_scheme", "fmgpu_search_ng21", "fmgpu_search_backtracking", "fmgpu_locate",
    "fmgpu_malloc", "fmgpu_free", "fmgpu_memcpy_h2d", "fmgpu_memcpy_d2h", "fmgpu_synchronize",
    "fmgpu_build_index", "fmgpu_built_free", "fmgpu_built_get", "fmgpu_index_accelerate", "fmgpu_index_accelerate_search",
    "fmgpu_index_accelerate_exact", "fmgpu_index_accelerate_locate", "fmgpu_hits_sort", "fmgpu_hits_pack16",
    "fmgpu_index_row_bits", "fmgpu_index_accelerate_lf", "fmgpu_search_exact_depth", "fmgpu_cursor_extend", "fmgpu_hits_pack24",
    "fmgpu_index_save", "fmgpu_index_load",
    "fmgpu_replicas_load", "fmgpu_replicas_destroy", "fmgpu_replicas_info", "fmgpu_replicas_search_exact", "fmgpu_replicas_search_scheme",
    "fmgpu_replicas_search_ng21", "fmgpu_replicas_locate",
    "fmgpu_set_option", "fmgpu_get_option", "fmgpu_index_formats", "fmgpu_index_clone", "fmgpu_replicas_peer_copies",
]

# fmgpu_option (include/fmgpu.h) and the defaults the library starts with
OPTIONS = {"pair_table": 0, "dense_dna": 1, "symbol_planes": 2, "expand_dna": 3, "lf_table": 4, "fused_locate": 5, "heavy_first": 6,
           "force_wide": 7, "kernel_select": 8, "fail_scratch": 9, "bucket_rows": 10, "suffix_sorter": 11}
OPTION_DEFAULTS = {"pair_table": 1, "dense_dna": 1, "symbol_planes": 1, "expand_dna": 1, "lf_table": 1, "fused_locate": 1, "heavy_first": 1,
                   "force_wide": 0, "kernel_select": 0, "fail_scratch": 0, "bucket_rows": 0, "suffix_sorter": 0}
# FMGPU_SEL_* bits of the kernel_select option
SEL_GENERAL_DFS, SEL_NO_PREFIX_TABLE, SEL_NO_LF3, SEL_NO_LF_GENERAL, SEL_NO_WALK_TABLE, SEL_NO_LENGTH_BUCKETS = 2, 4, 8, 16, 32, 64
SEL_EXACT_ON_TREE, SEL_EXACT_ONE_SYMBOL, SEL_LOCATE_PER_LANE, SEL_NO_SHARING, SEL_NO_EXACT_LUT, SEL_LEAN_FORMAT_A, SEL_NO_LEAN = 1 << 21, 1 << 22, 1 << 23, 1 << 24, 1 << 25, 1 << 29, 1 << 30
SEL_NO_BOARD = 1 << 26
# fmgpu_index_formats bits: what a handle holds beside (or as) the layout it was given
FMT_BLOCKS, FMT_PAIRS, FMT_DENSE, FMT_PLANES, FMT_TREE, FMT_REFERENCE, FMT_LF, FMT_KSTEP, FMT_INTERVALS, FMT_WALK, FMT_PREFIX, FMT_LOCATE, FMT_FUSED = (1 << k for k in range(13))

_lib = None


def _one_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so / libhsa-runtime64.so (same sonames as /opt/rocm's).  A process that loads
    libfmgpu.so first binds the system runtime, and a later `import torch` brings the bundled copy in beside it: two HIP runtimes in one
    process, and the second one to initialise does not find the GPU (hipErrorNoDevice).  Tensors and streams are shared with torch, so the
    runtime has to be ONE: if torch is installed but not imported yet, its copies are loaded first and libfmgpu.so binds to them by soname
    (exactly what happens when torch is imported first).  FMGPU_SYSTEM_HIP=1 leaves the loader alone."""
    import importlib.util
    import sys
    if "torch" in sys.modules or os.environ.get("FMGPU_SYSTEM_HIP") == "1":
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    libdir = os.path.join(os.path.dirname(spec.origin), "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            try:
                C.CDLL(path, mode=C.RTLD_GLOBAL)
            except OSError:
                return


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing — build it with __graft_entry__.build() "
                          "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    _one_hip_runtime()
    L = C.CDLL(LIB_PATH)
    L.fmgpu_last_error.restype = C.c_char_p
    L.fmgpu_device_count.argtypes = [C.POINTER(C.c_int)]
    L.fmgpu_set_device.argtypes = [C.c_int]
    L.fmgpu_index_create.argtypes = [C.POINTER(IndexDesc), C.POINTER(C.c_void_p)]
    L.fmgpu_index_destroy.argtypes = [C.c_void_p]
    L.fmgpu_index_info.argtypes = [C.c_void_p, u64p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), u64p]
    L.fmgpu_string_query.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    L.fmgpu_search_exact.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                                     C.POINTER(Stats), C.c_void_p]
    L.fmgpu_search_scheme.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(Scheme), C.c_uint64,
                                      C.c_void_p, C.c_uint64, u64p, C.POINTER(Stats), C.c_void_p]
    if hasattr(L, "fmgpu_search_exact_packed"):
        L.fmgpu_search_exact_packed.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(Stats), C.c_void_p]
    if hasattr(L, "fmgpu_search_ng21"):
        L.fmgpu_search_ng21.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(ExpandedScheme), C.c_uint64,
                                        C.c_void_p, C.c_uint64, u64p, C.POINTER(Stats), C.c_void_p]
    L.fmgpu_search_backtracking.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64,
                                            C.c_void_p, C.c_uint64, u64p, C.POINTER(Stats), C.c_void_p]
    L.fmgpu_locate.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p,
                               C.POINTER(Stats), C.c_void_p]
    if hasattr(L, "fmgpu_index_accelerate"):
        L.fmgpu_index_accelerate.argtypes = [C.c_void_p, C.c_int32]
    if hasattr(L, "fmgpu_hits_pack16"):
        L.fmgpu_hits_pack16.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    L.fmgpu_hits_pack24.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    if hasattr(L, "fmgpu_hits_sort"):
        L.fmgpu_hits_sort.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
    if hasattr(L, "fmgpu_index_accelerate_locate"):
        L.fmgpu_index_accelerate_locate.argtypes = [C.c_void_p, C.c_int32]
    if hasattr(L, "fmgpu_index_accelerate_exact"):
        L.fmgpu_index_accelerate_exact.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
    if hasattr(L, "fmgpu_index_accelerate_search"):
        L.fmgpu_index_accelerate_search.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    L.fmgpu_index_row_bits.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
    L.fmgpu_index_accelerate_lf.argtypes = [C.c_void_p, C.c_int32]
    L.fmgpu_search_exact_depth.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    L.fmgpu_cursor_extend.argtypes = [C.c_void_p, C.c_int32, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.fmgpu_index_save.argtypes = [C.c_void_p, C.c_char_p, C.c_int32]
    L.fmgpu_index_load.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
    L.fmgpu_index_clone.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    if hasattr(L, "fmgpu_replicas_load"):
        L.fmgpu_replicas_load.argtypes = [C.c_char_p, C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_void_p)]
        L.fmgpu_replicas_destroy.argtypes = [C.c_void_p]
        L.fmgpu_replicas_peer_copies.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
        L.fmgpu_replicas_info.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_void_p)]
        L.fmgpu_replicas_search_exact.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.POINTER(Stats)]
        L.fmgpu_replicas_search_scheme.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(Scheme), C.c_uint64, C.c_void_p, C.c_uint64, u64p, C.POINTER(Stats)]
        L.fmgpu_replicas_search_ng21.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(ExpandedScheme), C.c_uint64, C.c_void_p, C.c_uint64, u64p, C.POINTER(Stats)]
        L.fmgpu_replicas_locate.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Stats)]
    L.fmgpu_set_option.argtypes = [C.c_int32, C.c_int64]
    L.fmgpu_get_option.argtypes = [C.c_int32, C.POINTER(C.c_int64)]
    L.fmgpu_index_formats.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
    L.fmgpu_malloc.argtypes = [C.POINTER(C.c_void_p), C.c_uint64]
    L.fmgpu_free.argtypes = [C.c_void_p]
    L.fmgpu_memcpy_h2d.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    L.fmgpu_memcpy_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    L.fmgpu_synchronize.argtypes = [C.c_void_p]
    if hasattr(L, "fmgpu_build_index"):
        L.fmgpu_build_index.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32, C.c_int32, C.c_uint64, C.c_int32,
                                        C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
        L.fmgpu_built_free.argtypes = [C.c_void_p]
        L.fmgpu_built_get.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), u64p]
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise FmgpuError(rc, lib().fmgpu_last_error().decode("utf-8", "replace"))


def set_option(name, value):
    """fmgpu_set_option by name (OPTIONS) or number"""
    check(lib().fmgpu_set_option(OPTIONS[name] if isinstance(name, str) else int(name), int(value)))


def get_option(name):
    v = C.c_int64()
    check(lib().fmgpu_get_option(OPTIONS[name] if isinstance(name, str) else int(name), C.byref(v)))
    return int(v.value)


def ptr(a):
    """numpy array / int device pointer / None -> void*"""
    if a is None:
        return None
    if isinstance(a, (int, np.integer)):
        return C.c_void_p(int(a))
    if hasattr(a, "ptr") and hasattr(a, "nbytes") and not hasattr(a, "ctypes"):   # DeviceBuffer or any (ptr, nbytes) device view
        return C.c_void_p(a.ptr)
    return C.c_void_p(a.ctypes.data)


class DeviceBuffer:
    """a chunk of HBM owned by the caller (queries / results that stay resident)"""

    def __init__(self, nbytes):
        p = C.c_void_p()
        check(lib().fmgpu_malloc(C.byref(p), nbytes))
        self.ptr, self.nbytes = p.value, nbytes

    @classmethod
    def from_array(cls, a):
        a = np.ascontiguousarray(a)
        b = cls(max(a.nbytes, 8))
        check(lib().fmgpu_memcpy_h2d(C.c_void_p(b.ptr), C.c_void_p(a.ctypes.data), a.nbytes))
        return b

    def to_array(self, dtype, count):
        out = np.empty(count, dtype=dtype)
        check(lib().fmgpu_memcpy_d2h(C.c_void_p(out.ctypes.data), C.c_void_p(self.ptr), out.nbytes))
        return out

    def free(self):
        if self.ptr:
            lib().fmgpu_free(C.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
