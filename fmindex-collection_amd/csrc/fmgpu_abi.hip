// fmgpu_abi.hip — the extern "C" surface of libfmgpu.so (include/fmgpu.h) and everything that does not depend on the row width:
// error handling, staging helpers, the per-thread call scratch, the construction by-products.  Every entry point that takes an index
// handle is routed to the 32-bit-row build (namespace fmgpu32) or the 64-bit-row build (fmgpu64) of the kernels by IndexHeader::wide.
#include "fmgpu_common.h"

#include <cstdio>

#include <atomic>
#include <cstdlib>
#include <map>
#include <mutex>
#include <memory>
#include <new>

namespace fmgpu {

static thread_local std::string g_last_error;

void set_error(const std::string& msg) { g_last_error = msg; }
int fail(int code, const std::string& msg) { g_last_error = msg; return code; }
int hip_fail(hipError_t e, const char* what) {
    g_last_error = std::string(what) + ": " + hipGetErrorString(e);
    (void)hipGetLastError();
    return e == hipErrorOutOfMemory ? FMGPU_ERR_NOMEM : FMGPU_ERR_HIP;
}
const char* last_error_cstr() { return g_last_error.c_str(); }

bool is_device_pointer(const void* p) {
    if (!p) return false;
    hipPointerAttribute_t a;
    std::memset(&a, 0, sizeof a);
    hipError_t e = hipPointerGetAttributes(&a, p);
    if (e != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

int Staged::in(const void* p, size_t nbytes, hipStream_t s) {
    stream = s; bytes = nbytes;
    if (nbytes == 0) { dev = nullptr; return 0; }
    if (is_device_pointer(p)) { dev = const_cast<void*>(p); return 0; }
    FM_HIP(hipMalloc(&dev, nbytes));
    owned = true;
    FM_HIP(hipMemcpyAsync(dev, p, nbytes, hipMemcpyHostToDevice, s));
    return 0;
}
int Staged::out(void* p, size_t nbytes, hipStream_t s) {
    stream = s; bytes = nbytes;
    if (nbytes == 0) { dev = nullptr; return 0; }
    if (is_device_pointer(p)) { dev = p; return 0; }
    FM_HIP(hipMalloc(&dev, nbytes));
    owned = true; writeback = true; host = p;
    return 0;
}
int Staged::finish() {
    if (writeback && bytes) {
        FM_HIP(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, stream));
        FM_HIP(hipStreamSynchronize(stream));
        writeback = false;
    }
    return 0;
}
Staged::~Staged() {
    if (owned && dev) { (void)hipStreamSynchronize(stream); (void)hipFree(dev); }
}

bool want_wide(uint64_t n) { return n >= kNarrowLimit || opt_on(FMGPU_OPT_FORCE_WIDE); }

// ---- library options (fmgpu_set_option): process-wide, read when a call starts / a handle is made
static const int64_t kOptionDefaults[FMGPU_OPT_COUNT_] = {1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
static std::atomic<int64_t> g_options[FMGPU_OPT_COUNT_];
static std::once_flag g_options_once;
static void options_init() {
    for (int i = 0; i < FMGPU_OPT_COUNT_; ++i) g_options[i].store(kOptionDefaults[i], std::memory_order_relaxed);
#ifdef FMGPU_DEV                                                     // (development builds only: the environment names the initial values)
    static const char* const names[FMGPU_OPT_COUNT_] = {"FMGPU_PAIRS", "FMGPU_DENSE_DNA", "FMGPU_FLAT", "FMGPU_SHADOW", "FMGPU_LF_TABLE", "FMGPU_FUSED_LOCATE", "FMGPU_HEAVY_FIRST",
                                                        "FMGPU_FORCE_WIDE", "FMGPU_DEV_FLAGS", "FMGPU_FAIL_SCRATCH", "FMGPU_BUCKET_ROWS", "FMGPU_SUFFIX_SORTER"};
    for (int i = 0; i < FMGPU_OPT_COUNT_; ++i) if (const char* e = getenv(names[i])) g_options[i].store(atoll(e), std::memory_order_relaxed);
#endif
}
int64_t opt(int option) {
    std::call_once(g_options_once, options_init);
    return option >= 0 && option < FMGPU_OPT_COUNT_ ? g_options[option].load(std::memory_order_relaxed) : 0;
}
static int set_opt(int option, int64_t value) {
    std::call_once(g_options_once, options_init);
    if (option < 0 || option >= FMGPU_OPT_COUNT_) return fail(FMGPU_ERR_INVALID, "unknown option " + std::to_string(option));
    if (option == FMGPU_OPT_KERNEL_SELECT && (value & ~(int64_t)FMGPU_SEL_ALL)) return fail(FMGPU_ERR_INVALID, "FMGPU_OPT_KERNEL_SELECT: bits outside FMGPU_SEL_ALL");
    if (option == FMGPU_OPT_BUCKET_ROWS && value < 0) return fail(FMGPU_ERR_INVALID, "FMGPU_OPT_BUCKET_ROWS: a number of rows");
    if (option == FMGPU_OPT_SUFFIX_SORTER && (value < 0 || value > 3)) return fail(FMGPU_ERR_INVALID, "FMGPU_OPT_SUFFIX_SORTER: 0 .. 3");
    g_options[option].store(value, std::memory_order_relaxed);
    return 0;
}

// ---- per-thread, per-device scratch -----------------------------------------------------------------------------------------------
// Built into a local object and published only when every allocation has succeeded: a failed hipMalloc (plausible next to 224 GB of
// tables) leaves nothing half-initialised behind, and the next call simply tries again.  Keyed by device: a host thread that alternates
// between handles on two devices re-uses both sets.  FMGPU_OPT_FAIL_SCRATCH = k (test hook) fails the k-th allocation of the next creation.
void CallScratch::drop() {
    for (void* p : {(void*)ctr, (void*)sink, (void*)len2, frames, dfs_ctr, order, board}) if (p) (void)hipFree(p);
    if (pinned) (void)hipHostFree(pinned);
    if (ev_a) (void)hipEventDestroy(ev_a);
    if (ev_b) (void)hipEventDestroy(ev_b);
    for (hipStream_t st : dfs_streams) if (st) (void)hipStreamDestroy(st);
    for (hipEvent_t ev : dfs_events) if (ev) (void)hipEventDestroy(ev);
    *this = CallScratch{};
}
struct ScratchSet {
    std::map<int, CallScratch> by_dev;
    ~ScratchSet() { for (auto& kv : by_dev) kv.second.drop(); }   // a host thread that ends returns everything it held (errors of a runtime that is already shutting down are ignored)
};
int call_scratch(CallScratch** out) {
    static thread_local ScratchSet set;
    int dev = 0;
    FM_HIP(hipGetDevice(&dev));
    auto it = set.by_dev.find(dev);
    if (it != set.by_dev.end()) { *out = &it->second; return 0; }
    CallScratch sc;
    const int inject = (int)opt(FMGPU_OPT_FAIL_SCRATCH);
    int step = 0;
    auto guard = [&](hipError_t e, const char* what) -> int {
        ++step;
        if (inject && step == inject) e = hipErrorOutOfMemory;
        if (e == hipSuccess) return 0;
        sc.drop();
        return hip_fail(e, what);
    };
    const size_t ctr_bytes = (size_t)kCounterStripes * kCounterKinds * 8;
    int rc;
    if ((rc = guard(hipMalloc((void**)&sc.ctr, ctr_bytes), "hipMalloc(call scratch: counters)"))) return rc;
    if ((rc = guard(hipMalloc((void**)&sc.sink, ctr_bytes), "hipMalloc(call scratch: sink)"))) return rc;
    if ((rc = guard(hipMalloc((void**)&sc.len2, (2 * 1024 + 1) * 8), "hipMalloc(call scratch: length reduction)"))) return rc;
    if ((rc = guard(hipHostMalloc((void**)&sc.pinned, (2 * 1024 + 1) * 8, hipHostMallocDefault), "hipHostMalloc(call scratch)"))) return rc;
    if ((rc = guard(hipMalloc(&sc.dfs_ctr, 256), "hipMalloc(call scratch: DFS counters)"))) return rc;
    if ((rc = guard(hipEventCreate(&sc.ev_a), "hipEventCreate"))) return rc;
    if ((rc = guard(hipEventCreate(&sc.ev_b), "hipEventCreate"))) return rc;
    *out = &(set.by_dev[dev] = sc);
    return 0;
}

}  // namespace fmgpu

namespace fmgpu32 { namespace api {
#include "fmgpu_api_decl.h"
} }
namespace fmgpu64 { namespace api {
#include "fmgpu_api_decl.h"
} }

using namespace fmgpu;

static inline const IndexHeader* header_of(fmgpu_index_t h) {
    const IndexHeader* x = reinterpret_cast<const IndexHeader*>(h);
    return (x && x->magic == kIndexMagic) ? x : nullptr;
}
#define ROUTE(h, call)                                                                   \
    do {                                                                                 \
        const IndexHeader* hd_ = header_of(h);                                           \
        if (!hd_) return fail(FMGPU_ERR_INVALID, "index handle is null or not a handle of this library"); \
        return hd_->wide ? fmgpu64::api::call : fmgpu32::api::call;                      \
    } while (0)

extern "C" {

int fmgpu_abi_version(void) { return FMGPU_ABI_VERSION; }
int fmgpu_set_option(int32_t option, int64_t value) { return set_opt(option, value); }
int fmgpu_get_option(int32_t option, int64_t* value) {
    if (!value) return fail(FMGPU_ERR_INVALID, "value is null");
    if (option < 0 || option >= FMGPU_OPT_COUNT_) return fail(FMGPU_ERR_INVALID, "unknown option " + std::to_string(option));
    *value = opt(option);
    return 0;
}
const char* fmgpu_last_error(void) { return last_error_cstr(); }

int fmgpu_device_count(int* count) {
    if (!count) return fail(FMGPU_ERR_INVALID, "count is null");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { (void)hipGetLastError(); *count = 0; return fail(FMGPU_ERR_NO_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e)); }
    *count = c;
    return 0;
}
int fmgpu_set_device(int device) { FM_HIP(hipSetDevice(device)); return 0; }

int fmgpu_index_create(const fmgpu_index_desc* desc, fmgpu_index_t* out) {
    if (!desc || !out) return fail(FMGPU_ERR_INVALID, "desc / out is null");
    *out = nullptr;
    if (desc->bwt.n >= kWideLimit) return fail(FMGPU_ERR_UNSUPPORTED, "this build indexes fewer than 2^40 rows per string");
    return want_wide(desc->bwt.n) ? fmgpu64::api::fmgpu_index_create(desc, out) : fmgpu32::api::fmgpu_index_create(desc, out);
}
int fmgpu_build_index(const uint8_t* seqs, const uint64_t* seq_off, uint64_t nseq, int32_t sigma, int32_t layout, uint64_t sampling_rate, int32_t bidirectional,
                      int32_t keep_host, fmgpu_index_t* out, fmgpu_built_t* built) {
    if (!out) return fail(FMGPU_ERR_INVALID, "out is null");
    *out = nullptr;
    if (built) *built = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { (void)hipGetLastError(); return fail(FMGPU_ERR_NO_DEVICE, "no HIP device visible — the product path has no CPU fallback"); }
    if (!seqs || !seq_off || nseq == 0) return fail(FMGPU_ERR_INVALID, "seqs / seq_off is null or nseq == 0");
    uint64_t ends[2] = {0, 0};                                   // rows = symbols + one delimiter per sequence
    if (is_device_pointer(seq_off)) {
        FM_HIP(hipMemcpy(&ends[0], seq_off, 8, hipMemcpyDeviceToHost));
        FM_HIP(hipMemcpy(&ends[1], seq_off + nseq, 8, hipMemcpyDeviceToHost));
    } else { ends[0] = seq_off[0]; ends[1] = seq_off[nseq]; }
    if (ends[1] < ends[0]) return fail(FMGPU_ERR_INVALID, "seq_off is not non-decreasing");
    const uint64_t n = ends[1] - ends[0] + nseq;
    if (n >= kWideLimit) return fail(FMGPU_ERR_UNSUPPORTED, "this build indexes fewer than 2^40 rows");
    return want_wide(n) ? fmgpu64::api::fmgpu_build_index(seqs, seq_off, nseq, sigma, layout, sampling_rate, bidirectional, keep_host, out, built)
                        : fmgpu32::api::fmgpu_build_index(seqs, seq_off, nseq, sigma, layout, sampling_rate, bidirectional, keep_host, out, built);
}
int fmgpu_built_free(fmgpu_built_t b) { delete reinterpret_cast<Built*>(b); return 0; }
int fmgpu_built_get(fmgpu_built_t b_, int32_t part, const void** ptr, uint64_t* bytes) {
    Built* b = reinterpret_cast<Built*>(b_);
    if (!b || !ptr || !bytes) return fail(FMGPU_ERR_INVALID, "null argument");
    if (part < 0 || (size_t)part >= b->part.size()) return fail(FMGPU_ERR_INVALID, "no such part");
    *ptr = b->part[part].data(); *bytes = b->part[part].size();
    return 0;
}

int fmgpu_index_save(fmgpu_index_t h, const char* path, int32_t include_tables) { ROUTE(h, fmgpu_index_save(h, path, include_tables)); }
int fmgpu_index_load(const char* path, fmgpu_index_t* out) {
    if (!path || !out) return fail(FMGPU_ERR_INVALID, "path / out is null");
    *out = nullptr;
    FILE* f = fopen(path, "rb");
    if (!f) return fail(FMGPU_ERR_INVALID, std::string("index file: cannot open ") + path);
    struct { char magic[8]; uint32_t version, abi, wide, endian_probe; uint64_t meta_bytes, meta_sum, reserved[3]; } fh;
    static_assert(sizeof fh == 64, "file header is 64 bytes");
    int rc = 0;
    if (fread(&fh, 1, sizeof fh, f) != sizeof fh) rc = fail(FMGPU_ERR_INVALID, "index file: truncated (header)");
    else if (std::memcmp(fh.magic, "FMGPUIDX", 8) != 0) rc = fail(FMGPU_ERR_INVALID, "index file: not an index file of this library (the reference's cereal files are not read: see INTEGRATION.md)");
    else if (fh.endian_probe != 0x01020304u) rc = fail(FMGPU_ERR_UNSUPPORTED, "index file: written on a machine of the other byte order");
    else if (fh.version != 1) rc = fail(FMGPU_ERR_UNSUPPORTED, "index file: format version " + std::to_string(fh.version) + " (this library reads version 1)");
    else if (fh.abi != FMGPU_ABI_VERSION) rc = fail(FMGPU_ERR_UNSUPPORTED, "index file: written by ABI version " + std::to_string(fh.abi) + " of the library, this is " + std::to_string(FMGPU_ABI_VERSION) + " (the device formats may differ: rebuild the index)");
    else if (fh.wide > 1) rc = fail(FMGPU_ERR_INVALID, "index file: bad row width");
    else rc = fh.wide ? fmgpu64::api::index_load(f, &fh, out) : fmgpu32::api::index_load(f, &fh, out);
    fclose(f);
    return rc;
}
int fmgpu_index_destroy(fmgpu_index_t h) { if (!h) return 0; ROUTE(h, fmgpu_index_destroy(h)); }
int fmgpu_index_clone(fmgpu_index_t h, fmgpu_index_t* out) { ROUTE(h, fmgpu_index_clone(h, out)); }
int fmgpu_index_info(fmgpu_index_t h, uint64_t* n, int32_t* sigma, int32_t* layout, int32_t* bidirectional, uint64_t* device_bytes) {
    ROUTE(h, fmgpu_index_info(h, n, sigma, layout, bidirectional, device_bytes));
}
int fmgpu_index_formats(fmgpu_index_t h, uint32_t* mask) { ROUTE(h, fmgpu_index_formats(h, mask)); }
int fmgpu_index_row_bits(fmgpu_index_t h, int32_t* bits) {
    const IndexHeader* hd = header_of(h);
    if (!hd || !bits) return fail(FMGPU_ERR_INVALID, "index handle / bits is null");
    *bits = hd->wide ? 64 : 32;
    return 0;
}
int fmgpu_index_accelerate(fmgpu_index_t h, int32_t kstep) { ROUTE(h, fmgpu_index_accelerate(h, kstep)); }
int fmgpu_index_accelerate_exact(fmgpu_index_t h, int32_t kstep, int32_t lut_len, int32_t walk) { ROUTE(h, fmgpu_index_accelerate_exact(h, kstep, lut_len, walk)); }
int fmgpu_index_accelerate_search(fmgpu_index_t h, int32_t prefix_len, int32_t walk) { ROUTE(h, fmgpu_index_accelerate_search(h, prefix_len, walk)); }
int fmgpu_index_accelerate_locate(fmgpu_index_t h, int32_t enable) { ROUTE(h, fmgpu_index_accelerate_locate(h, enable)); }
int fmgpu_index_accelerate_lf(fmgpu_index_t h, int32_t enable) { ROUTE(h, fmgpu_index_accelerate_lf(h, enable)); }
int fmgpu_string_query(fmgpu_index_t h, int which, const uint64_t* idx, const uint8_t* symb, const uint8_t* what, uint64_t count, uint64_t* out, void* stream) {
    ROUTE(h, fmgpu_string_query(h, which, idx, symb, what, count, out, stream));
}
int fmgpu_search_exact(fmgpu_index_t h, const uint8_t* qbuf, const uint64_t* qoff, uint64_t nq, uint64_t* out_lb, uint64_t* out_len, fmgpu_stats* stats, void* stream) {
    ROUTE(h, fmgpu_search_exact(h, qbuf, qoff, nq, out_lb, out_len, stats, stream));
}
int fmgpu_search_exact_packed(fmgpu_index_t h, const uint8_t* qbuf, const uint64_t* qoff, uint64_t nq, uint64_t* out_interval, fmgpu_stats* stats, void* stream) {
    ROUTE(h, fmgpu_search_exact_packed(h, qbuf, qoff, nq, out_interval, stats, stream));
}
int fmgpu_search_exact_depth(fmgpu_index_t h, const uint8_t* qbuf, const uint64_t* qoff, uint64_t nq, uint32_t* out_depth, void* stream) {
    ROUTE(h, fmgpu_search_exact_depth(h, qbuf, qoff, nq, out_depth, stream));
}
int fmgpu_search_scheme(fmgpu_index_t h, const uint8_t* qbuf, const uint64_t* qoff, uint64_t nq, const fmgpu_scheme* scheme, uint64_t max_hits_per_query,
                        fmgpu_hit* out, uint64_t capacity, uint64_t* out_count, fmgpu_stats* stats, void* stream) {
    ROUTE(h, fmgpu_search_scheme(h, qbuf, qoff, nq, scheme, max_hits_per_query, out, capacity, out_count, stats, stream));
}
int fmgpu_search_ng21(fmgpu_index_t h, const uint8_t* qbuf, const uint64_t* qoff, uint64_t nq, const fmgpu_expanded_scheme* scheme, uint64_t max_hits_per_query,
                      fmgpu_hit* out, uint64_t capacity, uint64_t* out_count, fmgpu_stats* stats, void* stream) {
    ROUTE(h, fmgpu_search_ng21(h, qbuf, qoff, nq, scheme, max_hits_per_query, out, capacity, out_count, stats, stream));
}
int fmgpu_search_backtracking(fmgpu_index_t h, const uint8_t* qbuf, const uint64_t* qoff, uint64_t nq, uint64_t max_errors, fmgpu_hit* out, uint64_t capacity,
                              uint64_t* out_count, fmgpu_stats* stats, void* stream) {
    ROUTE(h, fmgpu_search_backtracking(h, qbuf, qoff, nq, max_errors, out, capacity, out_count, stats, stream));
}
int fmgpu_locate(fmgpu_index_t h, const uint64_t* rows, uint64_t count, uint64_t* out_seq, uint64_t* out_pos, uint64_t* out_steps, fmgpu_stats* stats, void* stream) {
    ROUTE(h, fmgpu_locate(h, rows, count, out_seq, out_pos, out_steps, stats, stream));
}
int fmgpu_cursor_extend(fmgpu_index_t h, int32_t direction, uint64_t count, const uint64_t* lb, const uint64_t* lb_rev, const uint64_t* len, const uint8_t* symb,
                        uint64_t* out_lb, uint64_t* out_lb_rev, uint64_t* out_len, void* stream) {
    ROUTE(h, fmgpu_cursor_extend(h, direction, count, lb, lb_rev, len, symb, out_lb, out_lb_rev, out_len, stream));
}

int fmgpu_malloc(void** ptr, uint64_t bytes) { if (!ptr) return fail(FMGPU_ERR_INVALID, "ptr is null"); FM_HIP(hipMalloc(ptr, bytes ? bytes : 8)); return 0; }
int fmgpu_free(void* ptr) { if (ptr) FM_HIP(hipFree(ptr)); return 0; }
int fmgpu_memcpy_h2d(void* dst, const void* src, uint64_t bytes) { if (bytes) FM_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice)); return 0; }
int fmgpu_memcpy_d2h(void* dst, const void* src, uint64_t bytes) { if (bytes) FM_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost)); return 0; }
int fmgpu_synchronize(void* stream) { FM_HIP(hipStreamSynchronize((hipStream_t)stream)); return 0; }

}  // extern "C"

namespace fmgpu32 { namespace api {
int fmgpu_hits_pack16(const fmgpu_hit* hits, uint64_t count, uint64_t* out, void* stream);
int fmgpu_hits_pack24(const fmgpu_hit* hits, uint64_t count, uint64_t* out, void* stream);
int fmgpu_hits_sort(fmgpu_hit* hits, uint64_t count, void* stream);
} }
extern "C" {
int fmgpu_hits_pack16(const fmgpu_hit* hits, uint64_t count, uint64_t* out, void* stream) { return fmgpu32::api::fmgpu_hits_pack16(hits, count, out, stream); }
int fmgpu_hits_pack24(const fmgpu_hit* hits, uint64_t count, uint64_t* out, void* stream) { return fmgpu32::api::fmgpu_hits_pack24(hits, count, out, stream); }
int fmgpu_hits_sort(fmgpu_hit* hits, uint64_t count, void* stream) { return fmgpu32::api::fmgpu_hits_sort(hits, count, stream); }
}
