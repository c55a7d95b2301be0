// fmgpu_index.hip — error handling, staging helpers, index upload / re-layout, String_c batch queries.
#include "fmgpu_common.h"

#include <algorithm>
#include <cstdlib>
#include <memory>
#include <new>

namespace fmgpu {

static thread_local std::string g_last_error;

void set_error(const std::string& msg) { g_last_error = msg; }
int fail(int code, const std::string& msg) { g_last_error = msg; return code; }
int hip_fail(hipError_t e, const char* what) {
    g_last_error = std::string(what) + ": " + hipGetErrorString(e);
    (void)hipGetLastError();
    return FMGPU_ERR_HIP;
}

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

// ------------------------------------------------------------------ layout parameters (mirror of the reference's struct layouts)
struct RefLayout {
    int family;      // reference family: 0 IB, 1 IBP, 2 EPR, 3 EPRV2, 4 wavelet
    uint32_t bt, K, bits_off, stride, rows, bitct;
    uint64_t period;
};

static int bit_width_u(uint64_t v) { int r = 0; while (v) { ++r; v >>= 1; } return r; }

static int ref_layout(int layout, int sigma, RefLayout& L) {
    uint32_t align = 8;
    L.bitct = (uint32_t)bit_width_u((uint64_t)sigma - 1);
    switch (layout) {
    case FMGPU_IB8:      L.family = 0; L.bt = 1; break;
    case FMGPU_IB16:     L.family = 0; L.bt = 2; break;
    case FMGPU_IB32:     L.family = 0; L.bt = 4; break;
    case FMGPU_IB16A:    L.family = 0; L.bt = 2; align = 64; break;
    case FMGPU_IBP16:    L.family = 1; L.bt = 2; break;
    case FMGPU_EPR8:     L.family = 2; L.bt = 1; break;
    case FMGPU_EPR16:    L.family = 2; L.bt = 2; break;
    case FMGPU_EPR32:    L.family = 2; L.bt = 4; break;
    case FMGPU_EPRV2_8:  L.family = 3; L.bt = 1; break;
    case FMGPU_EPRV2_16: L.family = 3; L.bt = 2; break;
    case FMGPU_EPRV2_32: L.family = 3; L.bt = 4; break;
    case FMGPU_WAVELET:  L.family = 4; L.bt = 0; return 0;
    default: return -1;
    }
    uint64_t full = 1ull << (8 * L.bt);
    L.rows = 64; L.period = full;
    if (L.family <= 1) L.K = (uint32_t)sigma;
    else if (L.family == 3) L.K = L.bitct;
    else { L.K = 1; L.rows = 64 / L.bitct; L.period = (full / L.rows) * L.rows; }
    L.bits_off = (uint32_t)(((uint64_t)sigma * L.bt + 7) / 8 * 8);
    L.stride = (uint32_t)(((uint64_t)L.bits_off + 8ull * L.K + align - 1) / align * align);
    return 0;
}

// ------------------------------------------------------------------ Format A conversion kernel
// thread = (device block B, symbol c).  Reference row p lives at bit (p+1)&63 of block (p+1)>>6
// (string/InterleavedBitvector.h:64-94); prefix layout stores s[j] <= c (InterleavedBitvectorPrefix.h:86-100).
__global__ __launch_bounds__(256) void k_convert_ib(const uint8_t* __restrict__ raw, const uint64_t* __restrict__ super,
                                                    const idx_t* __restrict__ C, uint8_t* __restrict__ out,
                                                    uint64_t nblocks, uint32_t sigma, uint32_t bt, uint32_t bits_off,
                                                    uint32_t stride, uint64_t period, uint32_t bstride, int prefix) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nblocks * sigma) return;
    uint64_t B = t / sigma;
    uint32_t c = (uint32_t)(t % sigma);
    auto count_of = [&](uint64_t b, uint32_t s) -> uint64_t {
        const uint8_t* p = raw + b * stride + (uint64_t)s * bt;
        if (bt == 2) return *reinterpret_cast<const uint16_t*>(p);
        if (bt == 1) return *p;
        return *reinterpret_cast<const uint32_t*>(p);
    };
    auto word_of = [&](uint64_t b, uint32_t s) -> uint64_t {
        return *reinterpret_cast<const uint64_t*>(raw + b * stride + bits_off + 8ull * s);
    };
    uint64_t sb = (64ull * B) / period;
    uint64_t w = word_of(B, c), cnt = count_of(B, c) + super[sb * sigma + c];
    uint64_t wn = B + 1 < nblocks ? word_of(B + 1, c) : 0;
    if (prefix && c > 0) {   // cumulative planes -> exclusive planes
        uint64_t wl = word_of(B, c - 1);
        uint64_t wln = B + 1 < nblocks ? word_of(B + 1, c - 1) : 0;
        uint64_t bmask = bt == 2 ? 0xffffull : (bt == 1 ? 0xffull : 0xffffffffull);
        cnt = ((count_of(B, c) - count_of(B, c - 1)) & bmask) + super[sb * sigma + c] - super[sb * sigma + c - 1];
        w &= ~wl; wn &= ~wln;
    }
    uint64_t bits = (w >> 1) | ((wn & 1ull) << 63);
    uint32_t total = (uint32_t)(cnt + (w & 1ull)) + C[c];
    uint32_t* o = reinterpret_cast<uint32_t*>(out + B * bstride + 12ull * c);
    o[0] = total; o[1] = (uint32_t)bits; o[2] = (uint32_t)(bits >> 32);
}

static int upload(const void* host, size_t bytes, void** dev);

// EPRV3 / EPRV4 / EPRV5 / InterleavedEPRV7 -> Format A.  thread = (block B, symbol c): the symbol-match mask of the bit planes
// (EPRV3.h:55-68) becomes the entry's bitmap (position p <-> bit p & 63, as in Format A), the counters of every level that
// cover row 64B (EPRV3.h:205-213, EPRV4.h:128-142, EPRV5.h:126-139, InterleavedEPRV7.h:190-199) are summed into cnt.
struct HierView {
    const uint8_t* bits; uint32_t bits_stride;
    const uint8_t* lev[3]; uint32_t lev_w[3], lev_shift[3], lev_stride[3], lev_off[3]; int nlev;
    const uint64_t* super; uint32_t sshift;
};
__global__ __launch_bounds__(256) void k_convert_hier(HierView v, const idx_t* __restrict__ C, uint8_t* __restrict__ out, uint64_t nblocks, uint64_t n,
                                                      uint32_t sigma, uint32_t bitct, uint32_t bstride) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nblocks * sigma) return;
    uint64_t B = t / sigma, row = B * 64;
    uint32_t c = (uint32_t)(t % sigma);
    uint64_t m = ~0ull;
    for (uint32_t i = 0; i < bitct; ++i) {
        uint64_t w; memcpy(&w, v.bits + B * v.bits_stride + 8ull * i, 8);            // V7's packed structs are not 8-byte aligned
        m &= w ^ (0ull - (uint64_t)((~c >> i) & 1u));
    }
    if (n - row < 64) m &= (1ull << (n - row)) - 1ull;                                 // rows past the end read as symbol 0 in the planes
    uint64_t cnt = v.super[(row >> v.sshift) * sigma + c];
    for (int L = 0; L < v.nlev; ++L) {
        const uint8_t* p = v.lev[L] + (row >> v.lev_shift[L]) * v.lev_stride[L] + v.lev_off[L] + (uint64_t)c * v.lev_w[L];
        if (v.lev_w[L] == 1) cnt += *p;
        else if (v.lev_w[L] == 2) { uint16_t x; memcpy(&x, p, 2); cnt += x; }
        else { uint32_t x; memcpy(&x, p, 4); cnt += x; }
    }
    uint32_t* o = reinterpret_cast<uint32_t*>(out + B * bstride + 12ull * c);
    o[0] = (uint32_t)cnt + C[c]; o[1] = (uint32_t)m; o[2] = (uint32_t)(m >> 32);
}

static int create_hier(const fmgpu_string_desc& d, const idx_t* dC, DevString& s) {
    const uint32_t sigma = (uint32_t)d.sigma, bitct = (uint32_t)bit_width_u((uint64_t)sigma - 1);
    HierView v{};
    v.bits_stride = 8 * bitct;
    switch (d.layout) {
    case FMGPU_EPRV3_8: case FMGPU_EPRV3_16: case FMGPU_EPRV3_32: {
        uint32_t bt = d.layout == FMGPU_EPRV3_8 ? 1 : (d.layout == FMGPU_EPRV3_16 ? 2 : 4);
        v.nlev = 1; v.lev_w[0] = bt; v.lev_shift[0] = 6; v.sshift = 8 * bt; break;
    }
    case FMGPU_EPRV4: v.nlev = 3; v.lev_w[0] = 1; v.lev_w[1] = 2; v.lev_w[2] = 4; v.lev_shift[0] = 6; v.lev_shift[1] = 8; v.lev_shift[2] = 16; v.sshift = 32; break;
    default:          v.nlev = 2; v.lev_w[0] = 1; v.lev_w[1] = 2; v.lev_shift[0] = 6; v.lev_shift[1] = 8; v.sshift = 16; break;
    }
    const bool v7 = d.layout == FMGPU_IEPRV7;
    if (v7) v.bits_stride += sigma;
    const uint64_t nblocks = d.n / 64 + 1, nsuper = (d.n >> v.sshift) + 1;
    if (!d.blocks || !d.super_blocks) return fail(FMGPU_ERR_INVALID, "bits / super_blocks missing");
    if (d.blocks_bytes < nblocks * v.bits_stride) return fail(FMGPU_ERR_INVALID, "bits array too short for n rows (expected >= " + std::to_string(nblocks * v.bits_stride) + " bytes)");
    if (d.n_super_blocks < nsuper) return fail(FMGPU_ERR_INVALID, "too few super blocks for n rows");
    void* held[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    auto drop = [&] { for (void* p : held) if (p) (void)hipFree(p); };
    int rc = upload(d.blocks, nblocks * v.bits_stride, &held[0]); if (rc) { drop(); return rc; }
    rc = upload(d.super_blocks, nsuper * sigma * 8, &held[1]); if (rc) { drop(); return rc; }
    v.bits = (const uint8_t*)held[0]; v.super = (const uint64_t*)held[1];
    for (int L = 0; L < v.nlev; ++L) {
        v.lev_stride[L] = v.lev_w[L] * sigma; v.lev_off[L] = 0;
        if (L == 0 && v7) { v.lev[0] = v.bits; v.lev_stride[0] = v.bits_stride; v.lev_off[0] = 8 * bitct; continue; }   // level0 inside the packed struct
        const uint64_t need = ((d.n >> v.lev_shift[L]) + 1) * v.lev_stride[L];
        if (!d.levels[L] || d.level_bytes[L] < need) { drop(); return fail(FMGPU_ERR_INVALID, "counter level " + std::to_string(L) + " missing or too short for n rows"); }
        rc = upload(d.levels[L], need, &held[2 + L]); if (rc) { drop(); return rc; }
        v.lev[L] = (const uint8_t*)held[2 + L];
    }
    const uint32_t bstride = sigma <= 5 ? 64u : 12u * sigma;
    s.blk_bytes = nblocks * bstride + 64;
    hipError_t e = hipMalloc(&s.blk, s.blk_bytes);
    if (e != hipSuccess) { drop(); return hip_fail(e, "hipMalloc(format A)"); }
    (void)hipMemset(s.blk, 0, s.blk_bytes);
    const uint64_t threads = nblocks * sigma;
    k_convert_hier<<<dim3((unsigned)((threads + 255) / 256)), dim3(256)>>>(v, dC, (uint8_t*)s.blk, nblocks, d.n, sigma, bitct, bstride);
    e = hipDeviceSynchronize();
    drop();
    if (e != hipSuccess) return hip_fail(e, "k_convert_hier");
    s.family = FAM_A; s.bitct = (int)bitct;
    s.va = ViewA{(const uint8_t*)s.blk, bstride, sigma, dC};
    return 0;
}

// FlattenedBitvectors2L<sigma, l1_bits, 65536> -> Format A (FlattenedBitvectors2L.h:209-224): thread = (64-row block B, symbol c)
__global__ __launch_bounds__(256) void k_convert_fbv(const uint8_t* __restrict__ bits, const uint64_t* __restrict__ l0, const uint16_t* __restrict__ l1,
                                                     const idx_t* __restrict__ C, uint8_t* __restrict__ out, uint64_t nblocks, uint64_t n,
                                                     uint32_t sigma, uint32_t bitct, uint32_t l1_bits, uint32_t bstride) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nblocks * sigma) return;
    const uint64_t B = t / sigma, row = B * 64, blk = row / l1_bits, sb = row >> 16;
    const uint32_t c = (uint32_t)(t % sigma), w = (uint32_t)((row % l1_bits) / 64), sig1 = sigma + 1;
    const uint8_t* base = bits + blk * ((uint64_t)bitct * l1_bits / 8);
    auto have = [&](uint32_t word) -> uint64_t {
        uint64_t m = ~0ull;
        for (uint32_t i = 0; i < bitct; ++i) {
            uint64_t v = *reinterpret_cast<const uint64_t*>(base + (uint64_t)i * (l1_bits / 8) + 8ull * word);
            m &= v ^ (0ull - (uint64_t)((~c >> i) & 1u));
        }
        return m;
    };
    uint64_t cnt = l0[sb * sig1 + c + 1] - l0[sb * sig1 + c] + (uint64_t)l1[blk * sig1 + c + 1] - (uint64_t)l1[blk * sig1 + c];
    for (uint32_t j = 0; j < w; ++j) cnt += (uint64_t)__popcll(have(j));
    uint64_t m = have(w);
    if (n - row < 64) m &= (1ull << (n - row)) - 1ull;
    uint32_t* o = reinterpret_cast<uint32_t*>(out + B * bstride + 12ull * c);
    o[0] = (uint32_t)cnt + C[c]; o[1] = (uint32_t)m; o[2] = (uint32_t)(m >> 32);
}

static int create_fbv(const fmgpu_string_desc& d, const idx_t* dC, DevString& s) {
    const uint32_t sigma = (uint32_t)d.sigma, bitct = (uint32_t)bit_width_u((uint64_t)sigma - 1);
    const uint32_t l1_bits = d.layout == FMGPU_FBV_64_64K ? 64u : (d.layout == FMGPU_FBV_512_64K ? 512u : 2048u);
    const uint64_t nsuper = d.n / 65536 + 1, nl1 = nsuper * (65536 / l1_bits), stride = (uint64_t)bitct * l1_bits / 8, nblocks = d.n / 64 + 1;
    if (!d.blocks || !d.super_blocks || !d.levels[0]) return fail(FMGPU_ERR_INVALID, "bits / l0 (super_blocks) / l1 (levels[0]) missing");
    if (d.blocks_bytes < nl1 * stride) return fail(FMGPU_ERR_INVALID, "bits array too short for n rows (expected " + std::to_string(nl1 * stride) + " bytes)");
    if (d.n_super_blocks < nsuper) return fail(FMGPU_ERR_INVALID, "too few l0 blocks for n rows");
    if (d.level_bytes[0] < nl1 * (sigma + 1) * 2) return fail(FMGPU_ERR_INVALID, "l1 array too short for n rows");
    void *db = nullptr, *d0 = nullptr, *d1 = nullptr;
    auto drop = [&] { if (db) (void)hipFree(db); if (d0) (void)hipFree(d0); if (d1) (void)hipFree(d1); };
    int rc = upload(d.blocks, nl1 * stride, &db); if (rc) { drop(); return rc; }
    rc = upload(d.super_blocks, nsuper * (sigma + 1) * 8, &d0); if (rc) { drop(); return rc; }
    rc = upload(d.levels[0], nl1 * (sigma + 1) * 2, &d1); if (rc) { drop(); return rc; }
    const uint32_t bstride = sigma <= 5 ? 64u : 12u * sigma;
    s.blk_bytes = nblocks * bstride + 64;
    hipError_t e = hipMalloc(&s.blk, s.blk_bytes);
    if (e != hipSuccess) { drop(); return hip_fail(e, "hipMalloc(format A)"); }
    (void)hipMemset(s.blk, 0, s.blk_bytes);
    const uint64_t threads = nblocks * sigma;
    k_convert_fbv<<<dim3((unsigned)((threads + 255) / 256)), dim3(256)>>>((const uint8_t*)db, (const uint64_t*)d0, (const uint16_t*)d1, dC, (uint8_t*)s.blk,
                                                                            nblocks, d.n, sigma, bitct, l1_bits, bstride);
    e = hipDeviceSynchronize();
    drop();
    if (e != hipSuccess) return hip_fail(e, "k_convert_fbv");
    s.family = FAM_A; s.bitct = (int)bitct;
    s.va = ViewA{(const uint8_t*)s.blk, bstride, sigma, dC};
    return 0;
}

static int upload(const void* host, size_t bytes, void** dev) {
    *dev = nullptr;
    if (bytes == 0) bytes = 8;
    FM_HIP(hipMalloc(dev, bytes));
    if (host) FM_HIP(hipMemcpy(*dev, host, bytes, hipMemcpyDefault));
    return 0;
}

int on_handle_device(const Index* x) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return fail(FMGPU_ERR_NO_DEVICE, "no HIP device visible — the product path has no CPU fallback"); }
    if (dev != x->device) return fail(FMGPU_ERR_INVALID, "the handle lives on device " + std::to_string(x->device) + ", the calling thread's current device is " + std::to_string(dev));
    return 0;
}

void free_string(DevString& s) {
    if (s.blk) (void)hipFree(s.blk);
    if (s.aux) (void)hipFree(s.aux);
    if (s.lf_table) (void)hipFree(s.lf_table);
    if (s.kblk) (void)hipFree(s.kblk);
    if (s.walk3) (void)hipFree(s.walk3);
    if (s.shadow) (void)hipFree(s.shadow);
    s.shadow = nullptr; s.shadow_bytes = 0;
    if (s.slut) (void)hipFree(s.slut);
    if (s.walkj) (void)hipFree(s.walkj);
    if (s.walk2j) (void)hipFree(s.walk2j);
    s.walk2j = nullptr;
    s.slut = nullptr; s.walkj = nullptr; s.slut_len = 0; s.slut_entries = 0; s.walk_J = 0;
    s.blk = s.aux = nullptr; s.lf_table = nullptr; s.kblk = nullptr; s.kstep = s.kcodes = 0; s.kblk_bytes = 0; s.walk3 = nullptr;
}

template <class Occ>
__global__ __launch_bounds__(256) void k_lf_table(Occ occ, uint64_t n, idx_t* __restrict__ out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t c;
    out[i] = occ.lf_symbol((idx_t)i, c);
}

int build_lf_table(DevString& s, hipStream_t stream) {
    const char* e = getenv("FMGPU_LF_TABLE");
    if (e && atoi(e) == 0) return 0;
    if (s.n == 0) return 0;
    FM_HIP(hipMalloc((void**)&s.lf_table, s.n * sizeof(idx_t) + 16));   // (+16: k_scheme_fast reads 16 bytes at a row)
    dim3 grid((unsigned)((s.n + 255) / 256)), block(256);
    switch (s.family) {
    case FAM_A:
        if (s.sigma == 5) k_lf_table<OccA<5>><<<grid, block, 0, stream>>>(OccA<5>{s.va}, s.n, s.lf_table);
        else k_lf_table<OccA<0>><<<grid, block, 0, stream>>>(OccA<0>{s.va}, s.n, s.lf_table);
        break;
    case FAM_EPR:   k_lf_table<OccR<false>><<<grid, block, 0, stream>>>(OccR<false>{s.vr}, s.n, s.lf_table); break;
    case FAM_EPRV2: k_lf_table<OccR<true>><<<grid, block, 0, stream>>>(OccR<true>{s.vr}, s.n, s.lf_table); break;
    default:        k_lf_table<OccW><<<grid, block, 0, stream>>>(OccW{s.vw}, s.n, s.lf_table); break;
    }
    FM_HIP(hipGetLastError());
    FM_HIP(hipStreamSynchronize(stream));
    return 0;
}

static int create_string(const fmgpu_string_desc& d, const idx_t* dC, DevString& s) {
    if (d.sigma < 2 || d.sigma > 256) return fail(FMGPU_ERR_INVALID, "sigma must be in [2, 256]");
    if (d.n >= 0xffffffffull - 64) return fail(FMGPU_ERR_UNSUPPORTED, "this build indexes fewer than 2^32 - 64 rows per string");
    if (d.layout >= FMGPU_EPRV3_8 && d.layout <= FMGPU_IEPRV7) {
        s.layout = d.layout; s.sigma = d.sigma; s.n = d.n;
        return create_hier(d, dC, s);
    }
    if (d.layout >= FMGPU_FBV_64_64K && d.layout <= FMGPU_FBV_2048_64K) {
        s.layout = d.layout; s.sigma = d.sigma; s.n = d.n;
        return create_fbv(d, dC, s);
    }
    RefLayout L{};
    if (ref_layout(d.layout, d.sigma, L) != 0) return fail(FMGPU_ERR_INVALID, "unknown layout id");
    s.layout = d.layout; s.sigma = d.sigma; s.n = d.n; s.bitct = (int)L.bitct;
    const uint32_t sigma = (uint32_t)d.sigma;

    if (L.family <= 1) {   // InterleavedBitvector* / InterleavedBitvectorPrefix* -> Format A
        uint64_t nblocks = d.n / 64 + 1, nsuper = d.n / L.period + 1;
        if (!d.blocks || !d.super_blocks) return fail(FMGPU_ERR_INVALID, "blocks / super_blocks missing");
        if (d.blocks_bytes != nblocks * L.stride) return fail(FMGPU_ERR_INVALID, "blocks_bytes does not match n / layout (expected " + std::to_string(nblocks * L.stride) + ")");
        if (d.n_super_blocks != nsuper) return fail(FMGPU_ERR_INVALID, "n_super_blocks does not match n / layout");
        void *raw = nullptr, *sup = nullptr;
        int rc = upload(d.blocks, d.blocks_bytes, &raw); if (rc) return rc;
        rc = upload(d.super_blocks, nsuper * sigma * 8, &sup); if (rc) { (void)hipFree(raw); return rc; }
        uint32_t bstride = sigma <= 5 ? 64u : 12u * sigma;
        s.blk_bytes = nblocks * bstride + 64;
        hipError_t e = hipMalloc(&s.blk, s.blk_bytes);
        if (e != hipSuccess) { (void)hipFree(raw); (void)hipFree(sup); return hip_fail(e, "hipMalloc(format A)"); }
        (void)hipMemset(s.blk, 0, s.blk_bytes);
        uint64_t threads = nblocks * sigma;
        k_convert_ib<<<dim3((unsigned)((threads + 255) / 256)), dim3(256)>>>(
            (const uint8_t*)raw, (const uint64_t*)sup, dC, (uint8_t*)s.blk, nblocks, sigma, L.bt, L.bits_off, L.stride,
            L.period, bstride, L.family == 1 ? 1 : 0);
        e = hipDeviceSynchronize();
        (void)hipFree(raw); (void)hipFree(sup);
        if (e != hipSuccess) return hip_fail(e, "k_convert_ib");
        s.family = FAM_A;
        s.va = ViewA{(const uint8_t*)s.blk, bstride, sigma, dC};
        return 0;
    }
    if (L.family == 2 || L.family == 3) {   // EPR / EPRV2: reference layout verbatim
        if (!d.blocks || !d.super_blocks) return fail(FMGPU_ERR_INVALID, "blocks / super_blocks missing");
        if (d.blocks_bytes % L.stride != 0) return fail(FMGPU_ERR_INVALID, "blocks_bytes is not a multiple of sizeof(Block)");
        uint64_t need_blocks = d.n / L.rows + 1;
        if (d.blocks_bytes / L.stride < need_blocks) return fail(FMGPU_ERR_INVALID, "too few blocks for n rows");
        if (d.n_super_blocks < d.n / L.period + 1) return fail(FMGPU_ERR_INVALID, "too few super blocks for n rows");
        int rc = upload(d.blocks, d.blocks_bytes, &s.blk); if (rc) return rc;
        rc = upload(d.super_blocks, d.n_super_blocks * sigma * 8, &s.aux); if (rc) return rc;
        s.blk_bytes = d.blocks_bytes; s.aux_bytes = d.n_super_blocks * sigma * 8;
        s.family = L.family == 2 ? FAM_EPR : FAM_EPRV2;
        ViewR v{};
        v.blk = (const uint8_t*)s.blk; v.super = (const uint64_t*)s.aux; v.C = dC;
        v.stride = L.stride; v.bits_off = L.bits_off; v.bt = L.bt; v.sigma = sigma; v.bitct = L.bitct;
        v.rows = L.rows; v.period_shift = 8 * L.bt; v.period = (uint32_t)std::min<uint64_t>(L.period, 0xffffffffull);
        // InterleavedEPR.h:28-47
        uint64_t entries = 64 / L.bitct, cm = (1ull << L.bitct) - 1, mk = 1ull << L.bitct;
        for (uint64_t i = 0; i < entries; i += 2) { v.maskEven = (v.maskEven << (2 * L.bitct)) | cm; v.bitMask = (v.bitMask << (2 * L.bitct)) | mk; }
        s.vr = v;
        return 0;
    }
    // wavelet -> Format W
    uint64_t nnodes = 1; while (nnodes < sigma) nnodes <<= 1;
    if (!d.nodes || d.n_nodes != nnodes) return fail(FMGPU_ERR_INVALID, "wavelet needs bit_ceil(sigma) node descriptors");
    std::vector<uint32_t> base(nnodes, 0);
    uint64_t total_lines = 0;
    for (uint64_t k = 0; k < nnodes; ++k) {
        base[k] = (uint32_t)total_lines;
        total_lines += d.nodes[k].total_length / 384 + 1;
        if (d.nodes[k].n_bits < d.nodes[k].total_length / 64 + 1) return fail(FMGPU_ERR_INVALID, "wavelet node bits array too short");
        if (total_lines >= 0xffffffffull) return fail(FMGPU_ERR_UNSUPPORTED, "wavelet too large for 32-bit line offsets");
    }
    std::unique_ptr<uint64_t[]> lines(new (std::nothrow) uint64_t[total_lines * 8]());
    if (!lines) return fail(FMGPU_ERR_NOMEM, "host staging for wavelet lines");
    for (uint64_t k = 0; k < nnodes; ++k) {
        const fmgpu_wavelet_node& nd = d.nodes[k];
        uint64_t nl = nd.total_length / 384 + 1, nwords = nd.total_length / 64 + 1, ones = 0;
        for (uint64_t li = 0; li < nl; ++li) {
            uint64_t* Lp = lines.get() + (base[k] + li) * 8;
            Lp[0] = ones;
            uint64_t cum = 0, h1 = 0;
            for (uint64_t j = 0; j < 6; ++j) {
                uint64_t wi = li * 6 + j;
                uint64_t w = wi < nwords ? nd.bits[wi] : 0;
                if (j) h1 |= cum << (9 * (j - 1));
                Lp[2 + j] = w;
                cum += (uint64_t)__builtin_popcountll(w);
            }
            Lp[1] = h1;
            ones += cum;
        }
    }
    int rc = upload(lines.get(), total_lines * 64, &s.blk); if (rc) return rc;
    rc = upload(base.data(), nnodes * 4, &s.aux); if (rc) return rc;
    s.blk_bytes = total_lines * 64; s.aux_bytes = nnodes * 4;
    s.family = FAM_WAVELET;
    s.vw = ViewW{(const uint64_t*)s.blk, (const uint32_t*)s.aux, dC, sigma, L.bitct};
    return 0;
}

// ------------------------------------------------------------------ multi-symbol-step table
// context code of row j: walk K LF steps from j collecting the BWT symbols s_1 (immediately before the suffix), s_2, ...;
// w = s_K ... s_1 in text order, code = sum (s_t - 1) * R^(t-1) with R = sigma - 1; 255 if a delimiter is met.
template <class Occ>
__global__ __launch_bounds__(256) void k_kstep_codes(Occ occ, const idx_t* __restrict__ lf_table, uint64_t n, uint32_t K, uint32_t R, uint8_t* __restrict__ code) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    idx_t row = (idx_t)j;
    uint32_t c = 0, mul = 1; bool ok = true;
    for (uint32_t t = 0; t < K; ++t) {
        uint32_t s;
        row = occ.lf_symbol(row, s);
        if (s == 0) { ok = false; break; }
        c += (s - 1) * mul; mul *= R;
    }
    (void)lf_table;
    code[j] = ok ? (uint8_t)c : (uint8_t)255;
}
// one wave per 64-row block: plane bits by ballot; per-block counts into cnt[w * nblocks + B]
__global__ __launch_bounds__(256) void k_kstep_bits(const uint8_t* __restrict__ code, uint64_t n, uint64_t nblocks, uint32_t ncodes,
                                                    uint8_t* __restrict__ kblk, uint32_t* __restrict__ cnt) {
    uint64_t B = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    uint32_t lane = threadIdx.x & 63u;
    if (B >= nblocks) return;
    uint64_t row = B * 64 + lane;
    uint32_t s = row < n ? code[row] : 0xffffffffu;
    for (uint32_t c0 = 0; c0 < ncodes; c0 += 64) {
        uint64_t mine = 0;
        for (uint32_t c = c0; c < ncodes && c < c0 + 64; ++c) {
            uint64_t bits = __ballot(s == c);
            if (lane == c - c0) mine = bits;
        }
        uint32_t c = c0 + lane;
        if (c < ncodes) {
            uint32_t* o = reinterpret_cast<uint32_t*>(kblk + (B * ncodes + c) * 16ull);
            o[1] = (uint32_t)mine; o[2] = (uint32_t)(mine >> 32); o[3] = 0;
            cnt[(uint64_t)c * nblocks + B] = (uint32_t)__popcll(mine);
        }
    }
}
// C_k[w] = LF_k(0, w): the k single steps of the context from row 0, last symbol first
template <class Occ>
__global__ void k_kstep_base(Occ occ, uint32_t K, uint32_t R, uint32_t ncodes, idx_t* __restrict__ base) {
    uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= ncodes) return;
    idx_t i = 0; uint32_t rest = w;
    for (uint32_t t = 0; t < K; ++t) { uint32_t s = rest % R + 1; rest /= R; i = occ.lf(i, s); }
    base[w] = i;
}
__global__ __launch_bounds__(256) void k_kstep_counts(const uint32_t* __restrict__ cnt, const idx_t* __restrict__ base, uint64_t nblocks, uint32_t ncodes,
                                                      uint8_t* __restrict__ kblk) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nblocks * ncodes) return;
    uint64_t B = t % nblocks; uint32_t c = (uint32_t)(t / nblocks);
    *reinterpret_cast<uint32_t*>(kblk + (B * ncodes + c) * 16ull) = cnt[t] + base[c];
}

// ------------------------------------------------------------------ search accelerators (prefix table, walk table)
__global__ __launch_bounds__(256) void k_walk3(const idx_t* __restrict__ lf, uint64_t n, idx_t* __restrict__ out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    idx_t a = lf[i], b = lf[a], c = lf[b];
    out[3 * i] = a; out[3 * i + 1] = b; out[3 * i + 2] = c;
}
// suffix table for exact search: the interval of the L symbols c_0 (consumed first = the query's last symbol), c_1, ...
template <class Occ>
__global__ __launch_bounds__(256) void k_suffix_lut(Occ occ, uint64_t entries, uint32_t L, uint32_t R, idx_t n, uint2* __restrict__ lut) {
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < entries; w += (uint64_t)gridDim.x * blockDim.x) {   // (2^32 entries: more than one launch's threads)
        idx_t lb = 0, len = n;
        uint64_t rest = w;
        for (uint32_t t = 0; t < L && len != 0; ++t) {
            uint32_t c = (uint32_t)(rest % R) + 1; rest /= R;
            idx_t ra, rb;
            occ.lf2(lb, lb + len, c, ra, rb);
            lb = ra; len = rb - ra;
        }
        lut[w] = make_uint2(lb, len);
    }
}
// J LF steps from every row, remembering the symbols met
__global__ __launch_bounds__(256) void k_walkj(const idx_t* __restrict__ lf, const idx_t* __restrict__ C, uint32_t sigma, uint64_t n, uint32_t J, uint32_t bits,
                                               uint2* __restrict__ out) {
    __shared__ idx_t sC[257];
    for (uint32_t i = threadIdx.x; i <= sigma; i += blockDim.x) sC[i] = C[i];
    __syncthreads();
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    idx_t r = (idx_t)i;
    uint32_t code = 0; bool ok = true;
    for (uint32_t t = 0; t < J; ++t) {
        idx_t nr = lf[r];
        uint32_t lo = 0, hi = sigma;                   // symbol of the step: C[s] <= LF < C[s+1]
        while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (sC[mid] <= nr) lo = mid; else hi = mid; }
        if (lo == 0) { ok = false; break; }
        code |= (lo - 1u) << (bits * t);
        r = nr;
    }
    out[i] = ok ? make_uint2(r, code) : make_uint2(0xffffffffu, 0u);
}
// 2J steps = two J-step entries chained
__global__ __launch_bounds__(256) void k_walk2j(const uint2* __restrict__ wj, uint64_t n, uint32_t* __restrict__ out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint2 a = wj[i];
    uint32_t r = 0xffffffffu, c0 = 0, c1 = 0;
    if (a.x != 0xffffffffu) { const uint2 b = wj[a.x]; if (b.x != 0xffffffffu) { r = b.x; c0 = a.y; c1 = b.y; } }
    out[3 * i] = r; out[3 * i + 1] = c0; out[3 * i + 2] = c1;
}
// bidirectional interval of every string w of L symbols in [1, sigma): extendRight symbol by symbol (fmindex/BiFMIndexCursor.h:121-128)
template <class Occ>
__global__ __launch_bounds__(256) void k_prefix_lut(Occ rv, uint64_t entries, uint32_t L, uint32_t R, idx_t n, uint4* __restrict__ lut) {
    const uint32_t sigma = rv.sigma();
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < entries; w += (uint64_t)gridDim.x * blockDim.x) {   // (2^32 entries: more than one launch's threads)
        idx_t lb = 0, lbRev = 0, len = n;
        uint32_t used = 0;
        uint64_t rest = w;
        for (uint32_t t = 0; t < L && len != 0; ++t) {
            uint32_t c = (uint32_t)(rest % R) + 1; rest /= R;
            idx_t pre = 0, ra = 0, rb = 0;
            for (uint32_t d = 0; d <= c && d < sigma; ++d) {
                idx_t x, y;
                rv.lf2(lbRev, lbRev + len, d, x, y);
                if (d < c) pre += y - x; else { ra = x; rb = y; }
            }
            lb += pre; lbRev = ra; len = rb - ra;
            ++used;
        }
        lut[w] = make_uint4(lb, lbRev, len, used);
    }
}

}  // namespace fmgpu

#include <hipcub/hipcub.hpp>

namespace fmgpu {

template <class Occ>
static int accelerate_with(DevString& s, Occ occ, uint32_t K) {
    const uint32_t R = (uint32_t)s.sigma - 1;
    uint64_t nc = 1;
    for (uint32_t t = 0; t < K; ++t) { nc *= R; if (nc > 255) return fail(FMGPU_ERR_UNSUPPORTED, "(sigma-1)^kstep must be <= 255"); }
    const uint32_t ncodes = (uint32_t)nc;
    const uint64_t n = s.n, nblocks = n / 64 + 1;
    uint8_t* code = nullptr; uint32_t* cnt = nullptr; idx_t* base = nullptr; uint8_t* kblk = nullptr; void* tmp = nullptr;
    auto cleanup = [&]() { for (void* p : {(void*)code, (void*)cnt, (void*)base, tmp}) if (p) (void)hipFree(p); };
    hipError_t e;
    if ((e = hipMalloc((void**)&code, n + 64)) != hipSuccess || (e = hipMalloc((void**)&cnt, (size_t)ncodes * nblocks * 4)) != hipSuccess ||
        (e = hipMalloc((void**)&base, ncodes * sizeof(idx_t))) != hipSuccess || (e = hipMalloc((void**)&kblk, (size_t)nblocks * ncodes * 16 + 128)) != hipSuccess) {
        cleanup(); if (kblk) (void)hipFree(kblk); return hip_fail(e, "hipMalloc(k-step table)");
    }
    k_kstep_codes<Occ><<<dim3((unsigned)((n + 255) / 256)), 256>>>(occ, s.lf_table, n, K, R, code);
    k_kstep_bits<<<dim3((unsigned)((nblocks * 64 + 255) / 256)), 256>>>(code, n, nblocks, ncodes, kblk, cnt);
    k_kstep_base<Occ><<<dim3((ncodes + 63) / 64), 64>>>(occ, K, R, ncodes, base);
    size_t tb = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, tb, cnt, cnt, (size_t)nblocks);
    if ((e = hipMalloc(&tmp, tb ? tb : 8)) != hipSuccess) { cleanup(); (void)hipFree(kblk); return hip_fail(e, "hipMalloc(scan)"); }
    for (uint32_t c = 0; c < ncodes; ++c) {
        uint32_t* p = cnt + (uint64_t)c * nblocks;
        size_t b2 = tb;
        if ((e = hipcub::DeviceScan::ExclusiveSum(tmp, b2, p, p, (size_t)nblocks)) != hipSuccess) { cleanup(); (void)hipFree(kblk); return hip_fail(e, "ExclusiveSum"); }
    }
    k_kstep_counts<<<dim3((unsigned)((nblocks * ncodes + 255) / 256)), 256>>>(cnt, base, nblocks, ncodes, kblk);
    e = hipDeviceSynchronize();
    cleanup();
    if (e != hipSuccess) { (void)hipFree(kblk); return hip_fail(e, "k-step table kernels"); }
    if (s.kblk) (void)hipFree(s.kblk);
    s.kblk = kblk; s.kstep = K; s.kcodes = ncodes; s.kblk_bytes = (size_t)nblocks * ncodes * 16 + 128;
    return 0;
}

// ------------------------------------------------------------------ String_c batch kernel
template <class Occ>
__global__ __launch_bounds__(256) void k_string_query(Occ occ, const uint64_t* __restrict__ idx, const uint8_t* __restrict__ symb,
                                                      const uint8_t* __restrict__ what, uint64_t count, uint64_t* __restrict__ out) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    idx_t i = (idx_t)idx[t];
    uint32_t c = symb ? symb[t] : 0;
    uint32_t w = what ? what[t] : 0;
    uint64_t r;
    if (w == 0) r = occ.rank(i, c);
    else if (w == 1) r = occ.prefix_rank(i, c);
    else r = occ.symbol(i);
    out[t] = r;
}

}  // namespace fmgpu

using namespace fmgpu;

extern "C" {

int fmgpu_abi_version(void) { return FMGPU_ABI_VERSION; }
const char* fmgpu_last_error(void) { return g_last_error.c_str(); }

int fmgpu_device_count(int* count) {
    if (!count) return fail(FMGPU_ERR_INVALID, "count is null");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { (void)hipGetLastError(); *count = 0; return fail(FMGPU_ERR_NO_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e)); }
    *count = c;
    return 0;
}
int fmgpu_set_device(int device) { FM_HIP(hipSetDevice(device)); return 0; }

int fmgpu_index_destroy(fmgpu_index_t h) {
    Index* x = reinterpret_cast<Index*>(h);
    if (!x) return 0;
    free_string(x->bwt); free_string(x->rev);
    for (void* p : {(void*)x->dC, x->sa_l0, x->sa_l1, x->sa_bits, x->sa_f0, x->sa_f1, (void*)x->lut, (void*)x->loc_tab}) if (p) (void)hipFree(p);
    delete x;
    return 0;
}

int fmgpu_index_create(const fmgpu_index_desc* desc, fmgpu_index_t* out) {
    if (!desc || !out) return fail(FMGPU_ERR_INVALID, "desc / out is null");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { (void)hipGetLastError(); return fail(FMGPU_ERR_NO_DEVICE, "no HIP device visible — the product path has no CPU fallback"); }
    if (!desc->C) return fail(FMGPU_ERR_INVALID, "C is null");
    if (desc->bwt_rev && (desc->bwt_rev->n != desc->bwt.n || desc->bwt_rev->sigma != desc->bwt.sigma))
        return fail(FMGPU_ERR_INVALID, "bwt don't have the same size: " + std::to_string(desc->bwt.n) + " " + std::to_string(desc->bwt_rev->n));   // fmindex/BiFMIndex.h:48-50
    std::unique_ptr<Index> x(new (std::nothrow) Index());
    if (!x) return fail(FMGPU_ERR_NOMEM, "host allocation");
    (void)hipGetDevice(&x->device);
    const int sigma = desc->bwt.sigma;
    if (sigma < 2 || sigma > 256) return fail(FMGPU_ERR_INVALID, "sigma must be in [2, 256]");
    std::vector<idx_t> c32(sigma + 1);
    for (int i = 0; i <= sigma; ++i) {
        if (desc->C[i] > desc->bwt.n) return fail(FMGPU_ERR_INVALID, "C[] entry exceeds n");
        x->hC[i] = desc->C[i]; c32[i] = (idx_t)desc->C[i];
    }
    int rc = upload(c32.data(), (sigma + 1) * sizeof(idx_t), (void**)&x->dC);
    auto bail = [&](int code) { fmgpu_index_destroy(reinterpret_cast<fmgpu_index_t>(x.release())); return code; };
    if (rc) return bail(rc);
    rc = create_string(desc->bwt, x->dC, x->bwt); if (rc) return bail(rc);
    if (desc->bwt_rev) { rc = create_string(*desc->bwt_rev, x->dC, x->rev); if (rc) return bail(rc); x->bidirectional = true; }
    rc = build_lf_table(x->bwt, nullptr); if (rc) return bail(rc);
    if (x->bidirectional) { rc = build_lf_table(x->rev, nullptr); if (rc) return bail(rc); }
    x->device_bytes = x->bwt.blk_bytes + x->bwt.aux_bytes + x->rev.blk_bytes + x->rev.aux_bytes +
                      (x->bwt.lf_table ? x->bwt.n * sizeof(idx_t) : 0) + (x->rev.lf_table ? x->rev.n * sizeof(idx_t) : 0);
    if (const fmgpu_sparse_array_desc* sa = desc->annotated_array) {
        if (sa->n != desc->bwt.n) return bail(fail(FMGPU_ERR_INVALID, "annotated_array.n != bwt.n"));
        if (sa->n_l0 < sa->n / 65536 + 1 || sa->n_l1 < sa->n / 512 + 1 || sa->n_bit_words < (sa->n / 512 + 1) * 8)
            return bail(fail(FMGPU_ERR_INVALID, "annotated_array presence bitvector arrays too short"));
        for (int f = 0; f < 2; ++f)
            if (sa->field[f].bits == 0 || sa->field[f].bits > 64 || sa->field[f].common_divisor == 0)
                return bail(fail(FMGPU_ERR_INVALID, "annotated_array dense vector has bits == 0 or > 64"));
        rc = upload(sa->l0, sa->n_l0 * 8, &x->sa_l0); if (rc) return bail(rc);
        rc = upload(sa->l1, sa->n_l1 * 2, &x->sa_l1); if (rc) return bail(rc);
        rc = upload(sa->bits, sa->n_bit_words * 8, &x->sa_bits); if (rc) return bail(rc);
        // one spare word so that the two-word read of dense_access never leaves the buffer
        for (int f = 0; f < 2; ++f) {
            void** dst = f == 0 ? &x->sa_f0 : &x->sa_f1;
            size_t bytes = (sa->field[f].n_words + 1) * 8;
            hipError_t e = hipMalloc(dst, bytes);
            if (e != hipSuccess) return bail(hip_fail(e, "hipMalloc(sa field)"));
            (void)hipMemset(*dst, 0, bytes);
            if (sa->field[f].n_words) {
                e = hipMemcpy(*dst, sa->field[f].data, sa->field[f].n_words * 8, hipMemcpyDefault);
                if (e != hipSuccess) return bail(hip_fail(e, "hipMemcpy(sa field)"));
            }
            x->device_bytes += bytes;
        }
        x->device_bytes += sa->n_l0 * 8 + sa->n_l1 * 2 + sa->n_bit_words * 8;
        x->vsa = ViewSA{(const uint64_t*)x->sa_l0, (const uint16_t*)x->sa_l1, (const uint64_t*)x->sa_bits,
                        (const uint64_t*)x->sa_f0, (const uint64_t*)x->sa_f1,
                        sa->field[0].bits, sa->field[1].bits, sa->field[0].common_divisor, sa->field[1].common_divisor};
        x->has_sa = true;
    }
    *out = reinterpret_cast<fmgpu_index_t>(x.release());
    return 0;
}

int fmgpu_index_accelerate(fmgpu_index_t h, int32_t kstep) {
    Index* x = reinterpret_cast<Index*>(h);
    if (!x) return fail(FMGPU_ERR_INVALID, "index handle is null");
    if (int drc = on_handle_device(x)) return drc;
    DevString& s = x->bwt;
    if (kstep < 0 || kstep > 8) return fail(FMGPU_ERR_INVALID, "kstep must be in [0, 8]");
    x->device_bytes -= s.kblk_bytes;
    if (s.kblk) { (void)hipFree(s.kblk); s.kblk = nullptr; s.kstep = s.kcodes = 0; s.kblk_bytes = 0; }
    if (kstep == 0) {                                  // drop the Format A shadows as well
        for (DevString* t : {&x->bwt, &x->rev}) if (t->shadow) { (void)hipFree(t->shadow); x->device_bytes -= t->shadow_bytes; t->shadow = nullptr; t->shadow_bytes = 0; t->va = ViewA{}; }
        return 0;
    }
    if (s.n == 0) return 0;
    int rc = 0;
    for (DevString* t : {&x->bwt, &x->rev}) {          // EPR / Wavelet strings: searches read a Format A expansion from here on
        if (t->n == 0 || t->family == FAM_A || t->shadow) continue;
        if ((rc = build_format_a_shadow(*t, x->dC, nullptr))) return rc;
        x->device_bytes += t->shadow_bytes;
    }
    if (kstep <= 1) return 0;
    switch (s.search_family()) {
    case FAM_A:     rc = s.sigma == 5 ? accelerate_with(s, OccA<5>{s.va}, (uint32_t)kstep) : accelerate_with(s, OccA<0>{s.va}, (uint32_t)kstep); break;
    case FAM_EPR:   rc = accelerate_with(s, OccR<false>{s.vr}, (uint32_t)kstep); break;
    case FAM_EPRV2: rc = accelerate_with(s, OccR<true>{s.vr}, (uint32_t)kstep); break;
    default:        rc = accelerate_with(s, OccW{s.vw}, (uint32_t)kstep); break;
    }
    if (rc == 0) x->device_bytes += s.kblk_bytes;
    return rc;
}

int fmgpu_index_accelerate_exact(fmgpu_index_t h, int32_t kstep, int32_t lut_len, int32_t walk) {
    Index* x = reinterpret_cast<Index*>(h);
    if (!x) return fail(FMGPU_ERR_INVALID, "index handle is null");
    if (int drc = on_handle_device(x)) return drc;
    if (lut_len < 0 || lut_len > 32) return fail(FMGPU_ERR_INVALID, "lut_len must be in [0, 32]");
    int rc = fmgpu_index_accelerate(h, kstep);
    if (rc) return rc;
    DevString& s = x->bwt;
    const uint64_t n = s.n;
    if (s.slut) { (void)hipFree(s.slut); x->device_bytes -= s.slut_entries * 8; s.slut = nullptr; s.slut_len = 0; s.slut_entries = 0; }
    if (s.walkj) { (void)hipFree(s.walkj); x->device_bytes -= n * 8; s.walkj = nullptr; s.walk_J = 0; }
    if (s.walk2j) { (void)hipFree(s.walk2j); x->device_bytes -= n * 12; s.walk2j = nullptr; }
    if (n == 0) return 0;
    const uint32_t sigma = (uint32_t)s.sigma, R = sigma - 1;
    if (lut_len > 0) {
        uint64_t entries = 1;
        for (int t = 0; t < lut_len; ++t) { entries *= R; if (entries > (1ull << 32)) return fail(FMGPU_ERR_UNSUPPORTED, "suffix table would exceed 2^32 entries"); }
        FM_HIP(hipMalloc((void**)&s.slut, entries * 8));
        dim3 grid((unsigned)std::min<uint64_t>((entries + 255) / 256, 1u << 22)), block(256);
        switch (s.search_family()) {
        case FAM_A:
            if (s.sigma == 5) k_suffix_lut<OccA<5>><<<grid, block>>>(OccA<5>{s.va}, entries, (uint32_t)lut_len, R, (idx_t)n, s.slut);
            else k_suffix_lut<OccA<0>><<<grid, block>>>(OccA<0>{s.va}, entries, (uint32_t)lut_len, R, (idx_t)n, s.slut);
            break;
        case FAM_EPR:   k_suffix_lut<OccR<false>><<<grid, block>>>(OccR<false>{s.vr}, entries, (uint32_t)lut_len, R, (idx_t)n, s.slut); break;
        case FAM_EPRV2: k_suffix_lut<OccR<true>><<<grid, block>>>(OccR<true>{s.vr}, entries, (uint32_t)lut_len, R, (idx_t)n, s.slut); break;
        default:        k_suffix_lut<OccW><<<grid, block>>>(OccW{s.vw}, entries, (uint32_t)lut_len, R, (idx_t)n, s.slut); break;
        }
        FM_HIP(hipDeviceSynchronize());
        s.slut_len = (uint32_t)lut_len; s.slut_entries = entries;
        x->device_bytes += entries * 8;
    }
    if (walk) {
        if (!s.lf_table) return fail(FMGPU_ERR_INVALID, "the walk table needs the LF table (FMGPU_LF_TABLE=0 was set)");
        uint32_t bits = 1; while ((1u << bits) < R) ++bits;               // symbols 1 .. sigma-1 stored as 0 .. sigma-2
        const uint32_t J = 32u / bits;
        FM_HIP(hipMalloc((void**)&s.walkj, n * 8 + 16));
        k_walkj<<<dim3((unsigned)((n + 255) / 256)), 256>>>(s.lf_table, x->dC, sigma, n, J, bits, s.walkj);
        FM_HIP(hipDeviceSynchronize());
        s.walk_J = J; s.walk_bits = bits;
        x->device_bytes += n * 8;
        if (walk >= 2) {
            FM_HIP(hipMalloc((void**)&s.walk2j, n * 12 + 16));
            k_walk2j<<<dim3((unsigned)((n + 255) / 256)), 256>>>(s.walkj, n, s.walk2j);
            FM_HIP(hipDeviceSynchronize());
            x->device_bytes += n * 12;
        }
    }
    return 0;
}

int fmgpu_index_accelerate_search(fmgpu_index_t h, int32_t prefix_len, int32_t walk) {
    Index* x = reinterpret_cast<Index*>(h);
    if (!x) return fail(FMGPU_ERR_INVALID, "index handle is null");
    if (int drc = on_handle_device(x)) return drc;
    if (!x->bidirectional) return fail(FMGPU_ERR_INVALID, "search accelerators need a BiFMIndex");
    if (prefix_len < 0 || prefix_len > 32) return fail(FMGPU_ERR_INVALID, "prefix_len must be in [0, 32]");
    const uint64_t n = x->bwt.n;
    // drop what exists
    if (x->lut) { (void)hipFree(x->lut); x->device_bytes -= x->lut_entries * 16; x->lut = nullptr; x->lut_len = 0; x->lut_entries = 0; }
    for (DevString* s : {&x->bwt, &x->rev}) if (s->walk3) { (void)hipFree(s->walk3); s->walk3 = nullptr; x->device_bytes -= n * 12; }
    if (x->rev.walkj) { (void)hipFree(x->rev.walkj); x->rev.walkj = nullptr; x->rev.walk_J = 0; x->device_bytes -= n * 8; }
    if (n == 0) return 0;
    if (walk) {
        const uint32_t sigma = (uint32_t)x->bwt.sigma;
        uint32_t bits = 1; while ((1u << bits) < sigma - 1) ++bits;
        for (DevString* s : {&x->bwt, &x->rev}) {
            if (!s->lf_table) return fail(FMGPU_ERR_INVALID, "walk tables need the LF tables (FMGPU_LF_TABLE=0 was set)");
            if (walk & 1) {
                FM_HIP(hipMalloc((void**)&s->walk3, n * 12 + 16));
                k_walk3<<<dim3((unsigned)((n + 255) / 256)), 256>>>(s->lf_table, n, s->walk3);
                x->device_bytes += n * 12;
            }
            if ((walk & 2) && !s->walkj) {                          // (the forward one may exist already: fmgpu_index_accelerate_exact)
                FM_HIP(hipMalloc((void**)&s->walkj, n * 8 + 16));
                k_walkj<<<dim3((unsigned)((n + 255) / 256)), 256>>>(s->lf_table, x->dC, sigma, n, 32u / bits, bits, s->walkj);
                s->walk_J = 32u / bits; s->walk_bits = bits;
                x->device_bytes += n * 8;
            }
        }
        FM_HIP(hipDeviceSynchronize());
    }
    if (prefix_len > 0) {
        const uint32_t R = (uint32_t)x->bwt.sigma - 1;
        uint64_t entries = 1;
        for (int t = 0; t < prefix_len; ++t) { entries *= R; if (entries > (1ull << 32)) return fail(FMGPU_ERR_UNSUPPORTED, "prefix table would exceed 2^32 entries"); }
        FM_HIP(hipMalloc((void**)&x->lut, entries * 16));
        dim3 grid((unsigned)std::min<uint64_t>((entries + 255) / 256, 1u << 22)), block(256);
        const DevString& r = x->rev;
        switch (r.family) {
        case FAM_A:
            if (r.sigma == 5) k_prefix_lut<OccA<5>><<<grid, block>>>(OccA<5>{r.va}, entries, (uint32_t)prefix_len, R, (idx_t)n, x->lut);
            else k_prefix_lut<OccA<0>><<<grid, block>>>(OccA<0>{r.va}, entries, (uint32_t)prefix_len, R, (idx_t)n, x->lut);
            break;
        case FAM_EPR:   k_prefix_lut<OccR<false>><<<grid, block>>>(OccR<false>{r.vr}, entries, (uint32_t)prefix_len, R, (idx_t)n, x->lut); break;
        case FAM_EPRV2: k_prefix_lut<OccR<true>><<<grid, block>>>(OccR<true>{r.vr}, entries, (uint32_t)prefix_len, R, (idx_t)n, x->lut); break;
        default:        k_prefix_lut<OccW><<<grid, block>>>(OccW{r.vw}, entries, (uint32_t)prefix_len, R, (idx_t)n, x->lut); break;
        }
        FM_HIP(hipDeviceSynchronize());
        x->lut_len = (uint32_t)prefix_len; x->lut_entries = entries;
        x->device_bytes += entries * 16;
    }
    return 0;
}

int fmgpu_index_info(fmgpu_index_t h, uint64_t* n, int32_t* sigma, int32_t* layout, int32_t* bidirectional, uint64_t* device_bytes) {
    Index* x = reinterpret_cast<Index*>(h);
    if (!x) return fail(FMGPU_ERR_INVALID, "index handle is null");
    if (n) *n = x->bwt.n;
    if (sigma) *sigma = x->bwt.sigma;
    if (layout) *layout = x->bwt.layout;
    if (bidirectional) *bidirectional = x->bidirectional ? 1 : 0;
    if (device_bytes) *device_bytes = x->device_bytes;
    return 0;
}

int fmgpu_string_query(fmgpu_index_t h, int which, const uint64_t* idx, const uint8_t* symb, const uint8_t* what,
                       uint64_t count, uint64_t* out, void* stream_) {
    Index* x = reinterpret_cast<Index*>(h);
    if (!x) return fail(FMGPU_ERR_INVALID, "index handle is null");
    if (int drc = on_handle_device(x)) return drc;
    if (which != 0 && !(which == 1 && x->bidirectional)) return fail(FMGPU_ERR_INVALID, "which must be 0 (bwt) or 1 (bwtRev of a BiFMIndex)");
    if (count == 0) return 0;
    if (!idx || !out) return fail(FMGPU_ERR_INVALID, "idx / out is null");
    hipStream_t stream = (hipStream_t)stream_;
    const DevString& s = which ? x->rev : x->bwt;
    Staged sidx, ssym, swhat, sout;
    int rc;
    if ((rc = sidx.in(idx, count * 8, stream))) return rc;
    if ((rc = ssym.in(symb, symb ? count : 0, stream))) return rc;
    if ((rc = swhat.in(what, what ? count : 0, stream))) return rc;
    if ((rc = sout.out(out, count * 8, stream))) return rc;
    dim3 grid((unsigned)((count + 255) / 256)), block(256);
    auto a = (const uint64_t*)sidx.dev; auto b = (const uint8_t*)ssym.dev; auto c = (const uint8_t*)swhat.dev; auto o = (uint64_t*)sout.dev;
    switch (s.family) {
    case FAM_A:
        if (s.sigma == 5) k_string_query<OccA<5>><<<grid, block, 0, stream>>>(OccA<5>{s.va}, a, b, c, count, o);
        else k_string_query<OccA<0>><<<grid, block, 0, stream>>>(OccA<0>{s.va}, a, b, c, count, o);
        break;
    case FAM_EPR:     k_string_query<OccR<false>><<<grid, block, 0, stream>>>(OccR<false>{s.vr}, a, b, c, count, o); break;
    case FAM_EPRV2:   k_string_query<OccR<true>><<<grid, block, 0, stream>>>(OccR<true>{s.vr}, a, b, c, count, o); break;
    default:          k_string_query<OccW><<<grid, block, 0, stream>>>(OccW{s.vw}, a, b, c, count, o); break;
    }
    FM_HIP(hipGetLastError());
    return sout.finish();
}

int fmgpu_malloc(void** ptr, uint64_t bytes) { if (!ptr) return fail(FMGPU_ERR_INVALID, "ptr is null"); FM_HIP(hipMalloc(ptr, bytes ? bytes : 8)); return 0; }
int fmgpu_free(void* ptr) { if (ptr) FM_HIP(hipFree(ptr)); return 0; }
int fmgpu_memcpy_h2d(void* dst, const void* src, uint64_t bytes) { if (bytes) FM_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice)); return 0; }
int fmgpu_memcpy_d2h(void* dst, const void* src, uint64_t bytes) { if (bytes) FM_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost)); return 0; }
int fmgpu_synchronize(void* stream) { FM_HIP(hipStreamSynchronize((hipStream_t)stream)); return 0; }

}  // extern "C"
