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

static void describe_index(const Index* x, bool tables, SavedIndex& m, const void* sp[2][11]) {
    std::memset(&m, 0, sizeof(SavedIndex));
    m.n = x->bwt.n; m.sigma = x->bwt.sigma; m.bidirectional = x->bidirectional; m.has_sa = x->has_sa; m.wide = kWide ? 1 : 0;
    std::memcpy(m.hC, x->hC, sizeof m.hC);
    describe_string(x->bwt, tables, m.str[0], sp[0]);
    describe_string(x->rev, tables, m.str[1], sp[1]);
    m.dC_bytes = ((uint64_t)x->bwt.sigma + 1) * sizeof(idx_t);
    if (x->has_sa) {
        for (int k = 0; k < 5; ++k) m.sa_bytes[k] = x->sa_bytes[k];
        m.vsa = x->vsa; m.vsa.l0 = nullptr; m.vsa.l1 = nullptr; m.vsa.bits = nullptr; m.vsa.f0 = nullptr; m.vsa.f1 = nullptr;
    }
    if (tables && x->loc_tab) m.loc_bytes = x->bwt.n * 12 + 16;
    if (tables && x->lut) { m.lut_bytes = x->lut_entries * 16; m.lut_entries = x->lut_entries; m.lut_len = x->lut_len; }
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
    const void* sp[2][11];
    describe_index(x, tables, *meta, sp);

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

// ---- load (called by the extern "C" entry point once the header says which row width the file holds) and clone (another device of the same process)
static int read_exact(FILE* f, void* p, size_t bytes, const char* what) {
    if (fread(p, 1, bytes, f) != bytes) return fail(FMGPU_ERR_INVALID, std::string("index file: truncated (") + what + ")");
    return 0;
}

// The description block is checked against n, sigma and the layouts BEFORE anything is allocated: its checksum is not keyed, so a stale file, a file of a build with the
// same sizeof(SavedIndex) or a hand-edited one would otherwise yield a handle whose kernels read outside the uploaded arrays.  Every size must be what creation
// would have allocated for (n, sigma, layout) — or at least that, where creation takes the caller's array as it is.
static int bad_meta(const std::string& what) { return fail(FMGPU_ERR_INVALID, "index file: inconsistent description block (" + what + ")"); }
static bool power_of(uint64_t base, uint32_t exp, uint64_t want, uint64_t cap) {
    uint64_t v = 1;
    for (uint32_t t = 0; t < exp; ++t) { v *= base; if (v > cap) return false; }
    return v == want;
}
static int validate_string(const SavedString& o, const SavedIndex& m, int w) {
    const std::string tag = w ? "bwtRev: " : "bwt: ";
    const uint64_t n = m.n, nblocks = n / 64 + 1;
    const uint32_t sigma = (uint32_t)m.sigma, R = sigma - 1;
    if (o.sigma != m.sigma || o.n != n) return bad_meta(tag + "n / sigma differ from the index's");
    if (o.layout < 0 || o.layout > FMGPU_FBV_2048_64K || o.family < FAM_A || o.family > FAM_WAVELET) return bad_meta(tag + "layout / family");
    if (o.bytes[SEC_BLK] == 0) return bad_meta(tag + "no block array");
    auto check_va = [&](uint64_t blk_bytes, uint64_t sup_bytes) -> int {
        const uint32_t bstride = sigma <= 5 ? 64u : 12u * sigma;
        if (o.va.bstride != bstride || o.va.sigma != sigma || o.va.fused > 1u || (o.va.fused && sigma > 5)) return bad_meta(tag + "block table view");
        if (blk_bytes < nblocks * bstride) return bad_meta(tag + "block table shorter than n / 64 + 1 blocks");
        if (kWide && sup_bytes < ((n >> kSuperShift) + 1) * sigma * 8) return bad_meta(tag + "super-block table too short");
        return 0;
    };
    int rc;
    if (o.family == FAM_A) { if ((rc = check_va(o.bytes[SEC_BLK], o.bytes[SEC_SUP]))) return rc; }
    else if (o.has_shadow) { if ((rc = check_va(o.bytes[SEC_SHADOW], o.bytes[SEC_SHADOW_SUP]))) return rc; }
    if (!o.has_shadow && (o.bytes[SEC_SHADOW] || o.bytes[SEC_SHADOW_SUP])) return bad_meta(tag + "expansion arrays without the expansion");
    if (o.has_shadow && (o.family == FAM_A || !o.bytes[SEC_SHADOW])) return bad_meta(tag + "expansion flag");
    if (o.family == FAM_EPR || o.family == FAM_EPRV2) {
        const ViewR& v = o.vr;
        const bool v2 = o.family == FAM_EPRV2;
        if (v.sigma != sigma || v.bitct < 1 || v.bitct > 8 || (v.bt != 1 && v.bt != 2 && v.bt != 4) || v.rows < 1 || v.rows > 64 || v.stride < 8 || v.stride > 4096 ||
            (uint64_t)sigma * v.bt > v.bits_off || (uint64_t)v.bits_off + 8ull * (v2 ? v.bitct : 1u) > v.stride || (!v2 && (v.period < 1 || v.rows != 64u / v.bitct)) ||
            (v2 && (v.rows != 64 || v.period_shift != 8u * v.bt)))
            return bad_meta(tag + "EPR block view");
        if (o.bytes[SEC_BLK] < (n / v.rows + 1) * v.stride) return bad_meta(tag + "too few EPR blocks for n rows");
        const uint64_t nsup = v2 ? ((v.period_shift >= 32 ? (n >> 32) : (n >> v.period_shift)) + 1) : n / v.period + 1;
        if (o.bytes[SEC_AUX] < nsup * sigma * 8) return bad_meta(tag + "too few EPR super-blocks for n rows");
    }
    if (o.family == FAM_WAVELET) {
        const ViewM& v = o.vm;
        if (v.sigma != sigma || v.nlevels < 1 || v.nlevels > (uint32_t)kMaxLevelsM || v.nnodes < 1 || v.nnodes > (uint32_t)kMaxNodesM || v.bitct < 1 || v.bitct > 8) return bad_meta(tag + "wavelet tree view");
        uint32_t nodes_seen = 0, bits_seen = 0;
        for (uint32_t l = 0; l < v.nlevels; ++l) {
            const LevelM& L = v.lv[l];
            if (L.bits < 1 || L.bits > 3 || L.stride < (4u << L.bits) + 8u * L.bits || L.stride > 64 || L.first_node != nodes_seen || L.shift + L.bits + bits_seen != v.bitct) return bad_meta(tag + "wavelet level");
            nodes_seen += 1u << bits_seen; bits_seen += L.bits;
        }
        if (nodes_seen != v.nnodes || bits_seen != v.bitct || o.bytes[SEC_AUX] < (uint64_t)v.nnodes * 8) return bad_meta(tag + "wavelet node table");
        uint64_t least = 0;                                       // every level holds the n positions once, cut into blocks of 64 per node
        for (uint32_t l = 0; l < v.nlevels; ++l) least += (n / 64) * v.lv[l].stride;
        if (o.bytes[SEC_BLK] < least) return bad_meta(tag + "wavelet tree shorter than its levels");
        if (kWide && (o.vm_super_off >= o.bytes[SEC_SUP] || o.vm_super_off < ((uint64_t)v.nnodes * 4 + 63) / 64 * 64 || (o.bytes[SEC_SUP] - o.vm_super_off) % 64)) return bad_meta(tag + "wavelet super-block table");
    }
    if (o.bytes[SEC_LF] && o.bytes[SEC_LF] != n * sizeof(idx_t) + 16) return bad_meta(tag + "LF table size");
    if (o.bytes[SEC_WALK3] && (o.bytes[SEC_WALK3] != n * 12 + 16 || kWide)) return bad_meta(tag + "walk table size");
    if (o.bytes[SEC_KBLK] && (kWide || o.kstep < 2 || !power_of(R, o.kstep, o.kcodes, 255) || o.bytes[SEC_KBLK] < nblocks * o.kcodes * 16)) return bad_meta(tag + "multi-symbol-step table");
    if (!o.bytes[SEC_KBLK] && (o.kstep || o.kcodes)) return bad_meta(tag + "multi-symbol-step fields without the table");
    if (o.bytes[SEC_SLUT] && (o.slut_len < 1 || o.slut_len > 32 || !power_of(R, o.slut_len, o.slut_entries, 1ull << 32) || o.bytes[SEC_SLUT] != o.slut_entries * kSlutEntryBytes)) return bad_meta(tag + "interval table");
    if (!o.bytes[SEC_SLUT] && (o.slut_len || o.slut_entries)) return bad_meta(tag + "interval-table fields without the table");
    if (o.bytes[SEC_WALKJ]) {
        uint32_t bits = 1; while ((1u << bits) < R) ++bits;
        if (o.bytes[SEC_WALKJ] != n * kWalkEntryBytes + 16 || o.walk_bits != bits || o.walk_J != 32u / bits) return bad_meta(tag + "LF^J walk table");
    } else if (o.walk_J || o.walk_bits) return bad_meta(tag + "walk fields without the table");
    if (o.bytes[SEC_WALK2J] && (o.bytes[SEC_WALK2J] != n * kWalk2EntryBytes + 16 || !o.bytes[SEC_WALKJ])) return bad_meta(tag + "LF^2J walk table");
    return 0;
}
static int validate_meta(const SavedIndex& m) {
    if ((m.wide != 0) != kWide || m.sigma < 2 || m.sigma > 256 || m.str[0].n != m.n || m.n >= kWideLimit || (!kWide && m.n >= kNarrowLimit)) return bad_meta("n / sigma / row width");
    if ((m.bidirectional != 0 && m.bidirectional != 1) || (m.has_sa != 0 && m.has_sa != 1)) return bad_meta("flags");
    const uint64_t n = m.n; const uint32_t sigma = (uint32_t)m.sigma;
    for (uint32_t c = 0; c < sigma; ++c) if (m.hC[c] > m.hC[c + 1]) return bad_meta("C is not ascending");
    if (m.hC[0] != 0 || m.hC[sigma] != n) return bad_meta("C does not end at n");
    if (m.dC_bytes != ((uint64_t)sigma + 1) * sizeof(idx_t)) return bad_meta("size of C");
    int rc;
    if ((rc = validate_string(m.str[0], m, 0))) return rc;
    if (m.bidirectional) { if ((rc = validate_string(m.str[1], m, 1))) return rc; }
    else {
        for (int k = 0; k < 11; ++k) if (m.str[1].bytes[k]) return bad_meta("arrays of a bwtRev the index does not have");
        if (m.lut_bytes) return bad_meta("prefix table without a bwtRev");
    }
    if (m.has_sa) {
        if (m.sa_bytes[0] < (n / 65536 + 1) * 8 || m.sa_bytes[1] < (n / 512 + 1) * 2 || m.sa_bytes[2] < (n / 512 + 1) * 64 || m.sa_bytes[3] < 8 || m.sa_bytes[4] < 8 || m.sa_bytes[3] % 8 || m.sa_bytes[4] % 8)
            return bad_meta("sampled suffix array arrays too short");
        if (m.vsa.bits0 < 1 || m.vsa.bits0 > 64 || m.vsa.bits1 < 1 || m.vsa.bits1 > 64 || m.vsa.div0 == 0 || m.vsa.div1 == 0) return bad_meta("sampled suffix array fields");
    } else {
        for (int k = 0; k < 5; ++k) if (m.sa_bytes[k]) return bad_meta("sampled suffix array arrays without the array");
        if (m.loc_bytes) return bad_meta("locate table without a sampled suffix array");
        if (m.str[0].va.fused) return bad_meta("fused presence bits without a sampled suffix array");
    }
    if (m.loc_bytes && (kWide || m.loc_bytes != n * 12 + 16)) return bad_meta("locate table size");
    if (m.lut_bytes && (kWide || m.lut_len < 1 || m.lut_len > 16 || !power_of(sigma - 1, m.lut_len, m.lut_entries, 1ull << 32) || m.lut_bytes != m.lut_entries * 16)) return bad_meta("prefix table");
    if (!m.lut_bytes && (m.lut_len || m.lut_entries)) return bad_meta("prefix-table fields without the table");
    return 0;
}

// where the arrays of a new handle come from: the sections of a file (through a pinned buffer, checksummed), or the arrays of a handle on another device of this
// process (hipMemcpyPeer: over xGMI, no host copy — SURVEY 8e: "upload once to GPU0 then hipMemcpyPeer")
struct FileSource {
    FILE* f; uint8_t* host; uint32_t nsec = 0;
    int get(uint32_t want_id, uint64_t want_bytes, void** dev) {
        *dev = nullptr;
        if (want_bytes == 0) return 0;
        SectionHeader sh{};
        int rc = read_exact(f, &sh, sizeof sh, "section header");
        if (rc) return rc;
        if (sh.id != want_id || sh.bytes != want_bytes)
            return fail(FMGPU_ERR_INVALID, "index file: section " + std::to_string(want_id) + " expected with " + std::to_string(want_bytes) + " bytes, found section " +
                                           std::to_string(sh.id) + " with " + std::to_string(sh.bytes));
        DBuf d;
        if ((rc = d.alloc(want_bytes + 64))) return rc;           // (+ the slack every table carries behind its last entry)
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
        ++nsec;
        return 0;
    }
    int finish() {
        uint64_t trailer[2] = {0, 0};
        int rc = read_exact(f, trailer, sizeof trailer, "trailer");
        if (rc) return rc;
        if (trailer[0] != kTrailerMagic || trailer[1] != nsec) return fail(FMGPU_ERR_INVALID, "index file: trailer does not match (truncated or not an index file)");
        return 0;
    }
};
struct PeerSource {
    const void* ptr[2][11]; const void* other[8]; int src_device, dst_device;     // other: by SectionId - SEC_C
    int get(uint32_t id, uint64_t bytes, void** dev) {
        *dev = nullptr;
        if (bytes == 0) return 0;
        const void* from = id >= SEC_C ? other[id - SEC_C] : ptr[id / 32][id % 32];
        if (!from) return fail(FMGPU_ERR_INVALID, "index clone: array " + std::to_string(id) + " is missing on the source handle");
        DBuf d; int rc;
        if ((rc = d.alloc(bytes + 64))) return rc;
        FM_HIP(hipMemcpyPeer(d.p, dst_device, from, src_device, bytes));
        *dev = d.take();
        return 0;
    }
    int finish() { return 0; }
};

// the handle from its description and its arrays (Source::get hands over one device array per call, in file order), then everything that is derived rather than stored
template <class Source>
static int index_assemble(const SavedIndex& m, Source& src, fmgpu_index_t* out) {
    const SavedIndex* meta = &m;
    int rc;
    std::unique_ptr<Index> x(new (std::nothrow) Index());
    if (!x) return fail(FMGPU_ERR_NOMEM, "host allocation");
    (void)hipGetDevice(&x->hdr.device);
    auto bail = [&](int code) { api::fmgpu_index_destroy(reinterpret_cast<fmgpu_index_t>(x.release())); return code; };
    std::memcpy(x->hC, meta->hC, sizeof meta->hC);
    x->bidirectional = meta->bidirectional != 0;
    for (int w = 0; w < 2; ++w) {
        DevString& s = w ? x->rev : x->bwt;
        const SavedString& o = meta->str[w];
        s.layout = o.layout; s.family = o.family; s.sigma = o.sigma; s.bitct = o.bitct; s.n = o.n;
        void* p[11];
        for (uint32_t k = 0; k < 11; ++k) {
            if ((rc = src.get(k + 32u * w, o.bytes[k], &p[k]))) { for (uint32_t q = 0; q < k; ++q) if (p[q]) (void)hipFree(p[q]); return bail(rc); }
        }
        s.blk = p[SEC_BLK]; s.blk_bytes = o.bytes[SEC_BLK]; s.aux = p[SEC_AUX]; s.aux_bytes = o.bytes[SEC_AUX]; s.sup = p[SEC_SUP]; s.sup_bytes = o.bytes[SEC_SUP];
        s.lf_table = (idx_t*)p[SEC_LF]; s.kblk = (uint8_t*)p[SEC_KBLK]; s.kblk_bytes = o.bytes[SEC_KBLK]; s.kstep = o.kstep; s.kcodes = o.kcodes;
        s.walk3 = (idx_t*)p[SEC_WALK3]; s.slut = (uint2*)p[SEC_SLUT]; s.slut_len = o.slut_len; s.slut_entries = o.slut_entries;
        s.walkj = (uint2*)p[SEC_WALKJ]; s.walk_J = o.walk_J; s.walk_bits = o.walk_bits; s.walk2j = (uint32_t*)p[SEC_WALK2J];
        s.shadow = p[SEC_SHADOW]; s.shadow_sup = p[SEC_SHADOW_SUP]; s.shadow_sup_bytes = o.bytes[SEC_SHADOW_SUP]; s.shadow_bytes = o.bytes[SEC_SHADOW] + o.bytes[SEC_SHADOW_SUP];
        for (uint32_t k = 0; k < 11; ++k) x->device_bytes += o.bytes[k];
    }
    void* dC = nullptr;
    if ((rc = src.get(SEC_C, meta->dC_bytes, &dC))) return bail(rc);
    x->dC = (idx_t*)dC;
    if (meta->has_sa) {
        void** dst[5] = {&x->sa_l0, &x->sa_l1, &x->sa_bits, &x->sa_f0, &x->sa_f1};
        for (uint32_t k = 0; k < 5; ++k) {
            if ((rc = src.get(SEC_SA_L0 + k, meta->sa_bytes[k], dst[k]))) return bail(rc);
            x->sa_bytes[k] = meta->sa_bytes[k]; x->device_bytes += meta->sa_bytes[k];
        }
        x->vsa = meta->vsa;
        x->vsa.l0 = (const uint64_t*)x->sa_l0; x->vsa.l1 = (const uint16_t*)x->sa_l1; x->vsa.bits = (const uint64_t*)x->sa_bits;
        x->vsa.f0 = (const uint64_t*)x->sa_f0; x->vsa.f1 = (const uint64_t*)x->sa_f1;
        x->has_sa = true;
    }
    void* q = nullptr;
    if ((rc = src.get(SEC_LOC, meta->loc_bytes, &q))) return bail(rc);
    x->loc_tab = (uint32_t*)q; x->device_bytes += meta->loc_bytes;
    if ((rc = src.get(SEC_LUT, meta->lut_bytes, &q))) return bail(rc);
    x->lut = (uint4*)q; x->lut_len = meta->lut_len; x->lut_entries = meta->lut_entries; x->device_bytes += meta->lut_bytes;
    if ((rc = src.finish())) return bail(rc);
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
            // the node offsets are data the kernels index the tree with: every node's block array must start inside the tree
            std::vector<uint64_t> off(s.vm.nnodes);
            hipError_t e = hipMemcpy(off.data(), s.aux, off.size() * 8, hipMemcpyDeviceToHost);
            if (e != hipSuccess) return bail(hip_fail(e, "index load (node offsets)"));
            for (uint32_t l = 0; l < s.vm.nlevels; ++l) {
                const uint32_t first = s.vm.lv[l].first_node, last = l + 1 < s.vm.nlevels ? s.vm.lv[l + 1].first_node : s.vm.nnodes;
                for (uint32_t k = first; k < last; ++k)
                    if (off[k] + s.vm.lv[l].stride > s.blk_bytes || (k > 0 && off[k] < off[k - 1])) return bail(bad_meta("wavelet node offsets leave the tree"));
            }
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
    { hipError_t e = hipDeviceSynchronize(); if (e != hipSuccess) return bail(hip_fail(e, "index load")); }     // (an asynchronous fault of a builder surfaces here: nothing is handed out)
    *out = reinterpret_cast<fmgpu_index_t>(x.release());
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
    if ((rc = validate_meta(*meta))) return rc;
    PinnedBuf pin;
    if ((rc = pin.alloc(kChunk))) return rc;
    FileSource src{f, (uint8_t*)pin.p};
    return index_assemble(*meta, src, out);
}

// a copy of the handle on the calling thread's current device: every stored array travels device to device, the derived tables are rebuilt there
int fmgpu_index_clone(fmgpu_index_t h, fmgpu_index_t* out) {
    Index* x = reinterpret_cast<Index*>(h);
    if (!x || !out) return fail(FMGPU_ERR_INVALID, "index handle / out is null");
    *out = nullptr;
    int dst = 0;
    FM_HIP(hipGetDevice(&dst));
    const int srcd = x->hdr.device;
    if (dst != srcd) {
        int can = 0;
        hipError_t e = hipDeviceCanAccessPeer(&can, dst, srcd);
        if (e != hipSuccess) { (void)hipGetLastError(); can = 0; }
        if (can) { e = hipDeviceEnablePeerAccess(srcd, 0); if (e != hipSuccess) (void)hipGetLastError(); }     // (already enabled is fine; hipMemcpyPeer stages through the host where it is not possible)
    }
    auto meta = std::make_unique<SavedIndex>();
    PeerSource src{};
    describe_index(x, true, *meta, src.ptr);
    src.other[SEC_C - SEC_C] = x->dC;
    src.other[SEC_SA_L0 - SEC_C] = x->sa_l0; src.other[SEC_SA_L1 - SEC_C] = x->sa_l1; src.other[SEC_SA_BITS - SEC_C] = x->sa_bits;
    src.other[SEC_SA_F0 - SEC_C] = x->sa_f0; src.other[SEC_SA_F1 - SEC_C] = x->sa_f1;
    src.other[SEC_LOC - SEC_C] = x->loc_tab; src.other[SEC_LUT - SEC_C] = x->lut;
    src.src_device = srcd; src.dst_device = dst;
    int rc;
    if ((rc = validate_meta(*meta))) return rc;                   // (a handle describes itself consistently: a guard for the code above, not for the caller)
    return index_assemble(*meta, src, out);
}
}  // namespace api
}  // namespace FMGPU_NS
