// fmgpu_file.hip — the library's own index file: fmgpu_index_save / fmgpu_index_load (include/fmgpu.h).
//
// The reference persists an index with cereal (fmindex/diskStorage.h:12-27: saveIndex / loadIndex over each struct's serialize()); its byte format
// for the mmser members cannot be pinned without a reference-written file, so this is NOT a reader of that format.  It is what SURVEY 5 asks for
// instead: a flat file — header, a POD description of the handle, then every device array as it sits in HBM — so that a process start costs one
// read and one copy per array instead of a suffix sort and 136-224 GB of table construction.
//
//   FileHeader   64 bytes: magic "FMGPUIDX", format version, ABI version, row width, byte order probe, size + checksum of the meta block
//   SavedIndex   POD: n, sigma, C, sampled-SA parameters, table inventory; two SavedString (bwt, bwtRev): layout, family, the view structs with
//                their pointers zeroed, the byte size of every array
//   sections     for every array present: { id, bytes, checksum } + payload, padded to 8 bytes
//   trailer      magic again + number of sections (a truncated file is recognised before anything is handed out)
// Checksums: 64-bit multiplicative hash over the payload's 64-bit words.  Load verifies everything; any mismatch, short read, unknown version or
// wrong row width is an error code, never a partly loaded handle.
#include "fmgpu_common.h"

#include <cstdio>
#include <memory>

namespace FMGPU_NS {

constexpr char kFileMagic[8] = {'F', 'M', 'G', 'P', 'U', 'I', 'D', 'X'};
constexpr uint32_t kFileVersion = 1;
constexpr uint64_t kTrailerMagic = 0x58444955504d4746ull ^ 0xffffffffffffffffull;
constexpr size_t kChunk = (size_t)64 << 20;

struct FileHeader {
    char magic[8]; uint32_t version, abi, wide, endian_probe;
    uint64_t meta_bytes, meta_sum, reserved[3];
};
static_assert(sizeof(FileHeader) == 64, "file header is 64 bytes");

enum SectionId : uint32_t {      // + 32 for the arrays of bwtRev
    SEC_BLK = 0, SEC_AUX, SEC_SUP, SEC_LF, SEC_KBLK, SEC_WALK3, SEC_SLUT, SEC_WALKJ, SEC_WALK2J, SEC_SHADOW, SEC_SHADOW_SUP,
    SEC_C = 64, SEC_SA_L0, SEC_SA_L1, SEC_SA_BITS, SEC_SA_F0, SEC_SA_F1, SEC_LOC, SEC_LUT
};
struct SectionHeader { uint32_t id, reserved; uint64_t bytes, sum; };

struct SavedString {
    int32_t layout, family, sigma, bitct;
    uint64_t n;
    uint64_t bytes[11];           // by SectionId (0 = absent)
    ViewA va; ViewR vr; ViewM vm; // pointers zeroed
    uint64_t vm_super_off;        // ViewM::super inside `sup` (wide rows)
    uint32_t kstep, kcodes, slut_len, walk_J, walk_bits, has_shadow;
    uint64_t slut_entries;
};
struct SavedIndex {
    uint64_t n; int32_t sigma, bidirectional, has_sa, wide;
    uint64_t hC[258];
    uint64_t sa_bytes[5];
    ViewSA vsa;                   // pointers zeroed
    uint64_t dC_bytes, loc_bytes, lut_bytes, lut_entries; uint32_t lut_len, reserved;
    SavedString str[2];
};

static uint64_t mix_words(uint64_t h, const uint8_t* p, size_t bytes) {
    size_t i = 0;
    for (; i + 8 <= bytes; i += 8) { uint64_t w; std::memcpy(&w, p + i, 8); h = (h ^ w) * 0x9e3779b97f4a7c15ull; h ^= h >> 29; }
    if (i < bytes) { uint64_t w = 0; std::memcpy(&w, p + i, bytes - i); h = (h ^ w) * 0x9e3779b97f4a7c15ull; h ^= h >> 29; }
    return h;
}

struct PinnedBuf {
    void* p = nullptr;
    int alloc(size_t b) { hipError_t e = hipHostMalloc(&p, b, hipHostMallocDefault); if (e != hipSuccess) { p = nullptr; return hip_fail(e, "hipHostMalloc(index file buffer)"); } return 0; }
    ~PinnedBuf() { if (p) (void)hipHostFree(p); }
};
struct FileCloser { void operator()(FILE* f) const { if (f) fclose(f); } };
using File = std::unique_ptr<FILE, FileCloser>;

// ---- save
static int write_section(FILE* f, uint32_t id, const void* dev, uint64_t bytes, uint8_t* host, uint32_t* nsec) {
    if (!dev || bytes == 0) return 0;
    // checksum first (one pass over the device array through the pinned buffer), then the payload (a second pass): the header precedes its payload
    // and the file is written strictly forward, so that it can go to a pipe
    uint64_t sum = 0x243f6a8885a308d3ull ^ id;
    for (uint64_t at = 0; at < bytes; at += kChunk) {
        const size_t c = (size_t)std::min<uint64_t>(kChunk, bytes - at);
        FM_HIP(hipMemcpy(host, (const uint8_t*)dev + at, c, hipMemcpyDeviceToHost));
        sum = mix_words(sum, host, c);
    }
    SectionHeader sh{id, 0, bytes, sum};
    if (fwrite(&sh, sizeof sh, 1, f) != 1) return fail(FMGPU_ERR_INVALID, "index file: write failed (section header)");
    for (uint64_t at = 0; at < bytes; at += kChunk) {
        const size_t c = (size_t)std::min<uint64_t>(kChunk, bytes - at);
        FM_HIP(hipMemcpy(host, (const uint8_t*)dev + at, c, hipMemcpyDeviceToHost));
        if (fwrite(host, 1, c, f) != c) return fail(FMGPU_ERR_INVALID, "index file: write failed (disk full?)");
    }
    const uint64_t pad = (8 - bytes % 8) % 8, zero = 0;
    if (pad && fwrite(&zero, 1, pad, f) != pad) return fail(FMGPU_ERR_INVALID, "index file: write failed");
    ++*nsec;
    return 0;
}

static void describe_string(const DevString& s, bool tables, SavedString& o, const void* ptr[11]) {
    std::memset(&o, 0, sizeof o);
    for (int k = 0; k < 11; ++k) ptr[k] = nullptr;
    o.layout = s.layout; o.family = s.family; o.sigma = s.sigma; o.bitct = s.bitct; o.n = s.n;
    if (s.n == 0 && !s.blk) return;
    const uint64_t n = s.n;
    auto put = [&](int id, const void* p, uint64_t b) { if (p && b) { ptr[id] = p; o.bytes[id] = b; } };
    put(SEC_BLK, s.blk, s.blk_bytes); put(SEC_AUX, s.aux, s.aux_bytes); put(SEC_SUP, s.sup, s.sup_bytes);
    o.vr = s.vr; o.vr.blk = nullptr; o.vr.super = nullptr; o.vr.C = nullptr;
    o.vm = s.vm; o.vm_super_off = (s.vm.super && s.sup) ? (uint64_t)((const uint8_t*)s.vm.super - (const uint8_t*)s.sup) : 0;
    o.vm.data = nullptr; o.vm.node_off = nullptr; o.vm.C = nullptr; o.vm.super = nullptr; o.vm.node_super = nullptr;
    const bool keep_shadow = tables && s.shadow;
    o.va = s.va; o.va.blk = nullptr; o.va.C = nullptr; o.va.super = nullptr;
    if (s.shadow && !keep_shadow) o.va = ViewA{};                 // (a Format A expansion that is not saved: the view described it)
    if (tables) {
        put(SEC_LF, s.lf_table, n * sizeof(idx_t) + 16);
        put(SEC_KBLK, s.kblk, s.kblk_bytes);
        put(SEC_WALK3, s.walk3, n * 12 + 16);
        put(SEC_SLUT, s.slut, s.slut_entries * kSlutEntryBytes);
        put(SEC_WALKJ, s.walkj, n * kWalkEntryBytes + 16);
        put(SEC_WALK2J, s.walk2j, n * kWalk2EntryBytes + 16);
        put(SEC_SHADOW, s.shadow, s.shadow_bytes - s.shadow_sup_bytes);
        put(SEC_SHADOW_SUP, s.shadow_sup, s.shadow_sup_bytes);
        o.kstep = s.kblk ? s.kstep : 0; o.kcodes = s.kblk ? s.kcodes : 0;
        o.slut_len = s.slut ? s.slut_len : 0; o.slut_entries = s.slut ? s.slut_entries : 0;
        o.walk_J = s.walkj ? s.walk_J : 0; o.walk_bits = s.walkj ? s.walk_bits : 0;
        o.has_shadow = s.shadow ? 1u : 0u;
    }
}

namespace api {
#include "fmgpu_api_decl.h"
int fmgpu_index_save(fmgpu_index_t h, const char* path, int32_t include_tables) {
    Index* x = reinterpret_cast<Index*>(h);
    if (!x || !path) return fail(FMGPU_ERR_INVALID, "index handle / path is null");
    if (int drc = on_handle_device(x)) return drc;
    FM_HIP(hipDeviceSynchronize());
    const bool tables = include_tables != 0;
    auto meta = std::make_unique<SavedIndex>();
    std::memset(meta.get(), 0, sizeof(SavedIndex));
    meta->n = x->bwt.n; meta->sigma = x->bwt.sigma; meta->bidirectional = x->bidirectional; meta->has_sa = x->has_sa; meta->wide = kWide ? 1 : 0;
    std::memcpy(meta->hC, x->hC, sizeof meta->hC);
    const void* sp[2][11];
    describe_string(x->bwt, tables, meta->str[0], sp[0]);
    describe_string(x->rev, tables, meta->str[1], sp[1]);
    meta->dC_bytes = ((uint64_t)x->bwt.sigma + 1) * sizeof(idx_t);
    if (x->has_sa) {
        for (int k = 0; k < 5; ++k) meta->sa_bytes[k] = x->sa_bytes[k];
        meta->vsa = x->vsa; meta->vsa.l0 = nullptr; meta->vsa.l1 = nullptr; meta->vsa.bits = nullptr; meta->vsa.f0 = nullptr; meta->vsa.f1 = nullptr;
    }
    if (tables && x->loc_tab) meta->loc_bytes = x->bwt.n * 12 + 16;
    if (tables && x->lut) { meta->lut_bytes = x->lut_entries * 16; meta->lut_entries = x->lut_entries; meta->lut_len = x->lut_len; }

    File f(fopen(path, "wb"));
    if (!f) return fail(FMGPU_ERR_INVALID, std::string("index file: cannot open for writing: ") + path);
    PinnedBuf pin; int rc;
    if ((rc = pin.alloc(kChunk))) return rc;
    uint8_t* host = (uint8_t*)pin.p;
    FileHeader fh{};
    std::memcpy(fh.magic, kFileMagic, 8); fh.version = kFileVersion; fh.abi = FMGPU_ABI_VERSION; fh.wide = kWide ? 1 : 0; fh.endian_probe = 0x01020304u;
    fh.meta_bytes = sizeof(SavedIndex); fh.meta_sum = mix_words(0x13198a2e03707344ull, (const uint8_t*)meta.get(), sizeof(SavedIndex));
    if (fwrite(&fh, sizeof fh, 1, f.get()) != 1 || fwrite(meta.get(), sizeof(SavedIndex), 1, f.get()) != 1) return fail(FMGPU_ERR_INVALID, "index file: write failed (header)");
    uint32_t nsec = 0;
    for (int w = 0; w < 2; ++w)
        for (uint32_t k = 0; k < 11; ++k)
            if ((rc = write_section(f.get(), k + 32u * w, sp[w][k], meta->str[w].bytes[k], host, &nsec))) return rc;
    if ((rc = write_section(f.get(), SEC_C, x->dC, meta->dC_bytes, host, &nsec))) return rc;
    const void* sa_ptr[5] = {x->sa_l0, x->sa_l1, x->sa_bits, x->sa_f0, x->sa_f1};
    if (x->has_sa) for (uint32_t k = 0; k < 5; ++k) if ((rc = write_section(f.get(), SEC_SA_L0 + k, sa_ptr[k], meta->sa_bytes[k], host, &nsec))) return rc;
    if ((rc = write_section(f.get(), SEC_LOC, meta->loc_bytes ? x->loc_tab : nullptr, meta->loc_bytes, host, &nsec))) return rc;
    if ((rc = write_section(f.get(), SEC_LUT, meta->lut_bytes ? x->lut : nullptr, meta->lut_bytes, host, &nsec))) return rc;
    const uint64_t trailer[2] = {kTrailerMagic, nsec};
    if (fwrite(trailer, sizeof trailer, 1, f.get()) != 1) return fail(FMGPU_ERR_INVALID, "index file: write failed (trailer)");
    FILE* raw = f.release();
    if (fclose(raw) != 0) return fail(FMGPU_ERR_INVALID, "index file: close failed (disk full?)");
    return 0;
}

// ---- load (called by the extern "C" entry point once the header says which row width the file holds)
static int read_exact(FILE* f, void* p, size_t bytes, const char* what) {
    if (fread(p, 1, bytes, f) != bytes) return fail(FMGPU_ERR_INVALID, std::string("index file: truncated (") + what + ")");
    return 0;
}
static int read_section(FILE* f, uint32_t want_id, uint64_t want_bytes, void** dev, uint8_t* host, uint32_t* nsec) {
    *dev = nullptr;
    if (want_bytes == 0) return 0;
    SectionHeader sh{};
    int rc = read_exact(f, &sh, sizeof sh, "section header");
    if (rc) return rc;
    if (sh.id != want_id || sh.bytes != want_bytes)
        return fail(FMGPU_ERR_INVALID, "index file: section " + std::to_string(want_id) + " expected with " + std::to_string(want_bytes) + " bytes, found section " +
                                       std::to_string(sh.id) + " with " + std::to_string(sh.bytes));
    DBuf d;
    if ((rc = d.alloc(want_bytes))) return rc;
    uint64_t sum = 0x243f6a8885a308d3ull ^ want_id;
    for (uint64_t at = 0; at < want_bytes; at += kChunk) {
        const size_t c = (size_t)std::min<uint64_t>(kChunk, want_bytes - at);
        if ((rc = read_exact(f, host, c, "section payload"))) return rc;
        sum = mix_words(sum, host, c);
        FM_HIP(hipMemcpy((uint8_t*)d.p + at, host, c, hipMemcpyHostToDevice));
    }
    uint64_t pad = (8 - want_bytes % 8) % 8, skip = 0;
    if (pad && (rc = read_exact(f, &skip, pad, "padding"))) return rc;
    if (sum != sh.sum) return fail(FMGPU_ERR_INVALID, "index file: checksum mismatch in section " + std::to_string(want_id));
    *dev = d.take();
    ++*nsec;
    return 0;
}

int index_load(FILE* f, const void* header, fmgpu_index_t* out) {
    const FileHeader& fh = *reinterpret_cast<const FileHeader*>(header);
    if (fh.meta_bytes != sizeof(SavedIndex)) return fail(FMGPU_ERR_UNSUPPORTED, "index file: written by another build of the library (description block of " + std::to_string(fh.meta_bytes) + " bytes)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { (void)hipGetLastError(); return fail(FMGPU_ERR_NO_DEVICE, "no HIP device visible — the product path has no CPU fallback"); }
    auto meta = std::make_unique<SavedIndex>();
    int rc;
    if ((rc = read_exact(f, meta.get(), sizeof(SavedIndex), "description block"))) return rc;
    if (mix_words(0x13198a2e03707344ull, (const uint8_t*)meta.get(), sizeof(SavedIndex)) != fh.meta_sum) return fail(FMGPU_ERR_INVALID, "index file: checksum mismatch in the description block");
    if ((meta->wide != 0) != kWide || meta->sigma < 2 || meta->sigma > 256 || meta->str[0].n != meta->n || meta->n >= kWideLimit)
        return fail(FMGPU_ERR_INVALID, "index file: inconsistent description block");
    std::unique_ptr<Index> x(new (std::nothrow) Index());
    if (!x) return fail(FMGPU_ERR_NOMEM, "host allocation");
    (void)hipGetDevice(&x->hdr.device);
    auto bail = [&](int code) { api::fmgpu_index_destroy(reinterpret_cast<fmgpu_index_t>(x.release())); return code; };
    PinnedBuf pin;
    if ((rc = pin.alloc(kChunk))) return bail(rc);
    uint8_t* host = (uint8_t*)pin.p;
    uint32_t nsec = 0;
    std::memcpy(x->hC, meta->hC, sizeof meta->hC);
    x->bidirectional = meta->bidirectional != 0;
    for (int w = 0; w < 2; ++w) {
        DevString& s = w ? x->rev : x->bwt;
        const SavedString& o = meta->str[w];
        s.layout = o.layout; s.family = o.family; s.sigma = o.sigma; s.bitct = o.bitct; s.n = o.n;
        void* p[11];
        for (uint32_t k = 0; k < 11; ++k) {
            if ((rc = read_section(f, k + 32u * w, o.bytes[k], &p[k], host, &nsec))) { for (uint32_t q = 0; q < k; ++q) if (p[q]) (void)hipFree(p[q]); return bail(rc); }
        }
        s.blk = p[SEC_BLK]; s.blk_bytes = o.bytes[SEC_BLK]; s.aux = p[SEC_AUX]; s.aux_bytes = o.bytes[SEC_AUX]; s.sup = p[SEC_SUP]; s.sup_bytes = o.bytes[SEC_SUP];
        s.lf_table = (idx_t*)p[SEC_LF]; s.kblk = (uint8_t*)p[SEC_KBLK]; s.kblk_bytes = o.bytes[SEC_KBLK]; s.kstep = o.kstep; s.kcodes = o.kcodes;
        s.walk3 = (idx_t*)p[SEC_WALK3]; s.slut = (uint2*)p[SEC_SLUT]; s.slut_len = o.slut_len; s.slut_entries = o.slut_entries;
        s.walkj = (uint2*)p[SEC_WALKJ]; s.walk_J = o.walk_J; s.walk_bits = o.walk_bits; s.walk2j = (uint32_t*)p[SEC_WALK2J];
        s.shadow = p[SEC_SHADOW]; s.shadow_sup = p[SEC_SHADOW_SUP]; s.shadow_sup_bytes = o.bytes[SEC_SHADOW_SUP]; s.shadow_bytes = o.bytes[SEC_SHADOW] + o.bytes[SEC_SHADOW_SUP];
        for (uint32_t k = 0; k < 11; ++k) x->device_bytes += o.bytes[k];
    }
    void* dC = nullptr;
    if ((rc = read_section(f, SEC_C, meta->dC_bytes, &dC, host, &nsec))) return bail(rc);
    x->dC = (idx_t*)dC;
    if (meta->has_sa) {
        void** dst[5] = {&x->sa_l0, &x->sa_l1, &x->sa_bits, &x->sa_f0, &x->sa_f1};
        for (uint32_t k = 0; k < 5; ++k) {
            if ((rc = read_section(f, SEC_SA_L0 + k, meta->sa_bytes[k], dst[k], host, &nsec))) return bail(rc);
            x->sa_bytes[k] = meta->sa_bytes[k]; x->device_bytes += meta->sa_bytes[k];
        }
        x->vsa = meta->vsa;
        x->vsa.l0 = (const uint64_t*)x->sa_l0; x->vsa.l1 = (const uint16_t*)x->sa_l1; x->vsa.bits = (const uint64_t*)x->sa_bits;
        x->vsa.f0 = (const uint64_t*)x->sa_f0; x->vsa.f1 = (const uint64_t*)x->sa_f1;
        x->has_sa = true;
    }
    void* q = nullptr;
    if ((rc = read_section(f, SEC_LOC, meta->loc_bytes, &q, host, &nsec))) return bail(rc);
    x->loc_tab = (uint32_t*)q; x->device_bytes += meta->loc_bytes;
    if ((rc = read_section(f, SEC_LUT, meta->lut_bytes, &q, host, &nsec))) return bail(rc);
    x->lut = (uint4*)q; x->lut_len = meta->lut_len; x->lut_entries = meta->lut_entries; x->device_bytes += meta->lut_bytes;
    uint64_t trailer[2] = {0, 0};
    if ((rc = read_exact(f, trailer, sizeof trailer, "trailer"))) return bail(rc);
    if (trailer[0] != kTrailerMagic || trailer[1] != nsec) return bail(fail(FMGPU_ERR_INVALID, "index file: trailer does not match (truncated or not an index file)"));
    // the views: the saved scalars with this process's device pointers
    for (int w = 0; w < 2; ++w) {
        DevString& s = w ? x->rev : x->bwt;
        const SavedString& o = meta->str[w];
        if (!s.blk) continue;
        if (s.family == FAM_A || s.shadow) {
            s.va = o.va; s.va.C = x->dC;
            s.va.blk = (const uint8_t*)(s.shadow ? s.shadow : s.blk);
            s.va.super = (const uint64_t*)(s.shadow ? s.shadow_sup : s.sup);
        }
        if (s.family == FAM_EPR || s.family == FAM_EPRV2) { s.vr = o.vr; s.vr.blk = (const uint8_t*)s.blk; s.vr.super = (const uint64_t*)s.aux; s.vr.C = x->dC; }
        if (s.family == FAM_WAVELET) {
            s.vm = o.vm; s.vm.data = (const uint8_t*)s.blk; s.vm.node_off = (const uint64_t*)s.aux; s.vm.C = x->dC;
            s.vm.node_super = kWide ? (const uint32_t*)s.sup : nullptr;
            s.vm.super = (kWide && s.sup) ? reinterpret_cast<const uint64_t*>((const uint8_t*)s.sup + o.vm_super_off) : nullptr;
        }
    }
    // derived data is rebuilt rather than stored: the Format A expansion of a sigma = 5 string of another layout (unless the file carried it with its tables) ...
    {
        const size_t had = x->bwt.shadow_bytes + x->rev.shadow_bytes;
        if ((rc = auto_shadow(x.get(), nullptr))) return bail(rc);
        x->device_bytes += x->bwt.shadow_bytes + x->rev.shadow_bytes - had;
        if ((rc = fuse_presence_bits(x.get(), nullptr))) return bail(rc);          // (no-op where the saved blocks carry the bits already)
    }
    // ... Format D (a few ms)
    if (x->bidirectional) {
        for (DevString* t : {&x->bwt, &x->rev}) if ((rc = build_dense_dna(*t, nullptr))) return bail(rc);
        if (x->bwt.dense && !x->rev.dense) { (void)hipFree(x->bwt.dense); (void)hipFree(x->bwt.dense_ex); x->bwt.dense = nullptr; x->bwt.dense_ex = nullptr; x->bwt.dense_bytes = 0; x->bwt.dense_nex = 0; }
        x->device_bytes += x->bwt.dense_bytes + x->rev.dense_bytes;
    }
    if ((rc = build_pair_table(x.get(), nullptr))) return bail(rc);       // ... and so are Formats P and S
    if ((rc = build_flat_table(x.get(), nullptr))) return bail(rc);
    FM_HIP(hipDeviceSynchronize());
    *out = reinterpret_cast<fmgpu_index_t>(x.release());
    return 0;
}
}  // namespace api
}  // namespace FMGPU_NS
