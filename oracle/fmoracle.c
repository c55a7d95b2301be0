/* oracle/fmoracle.c — CPU restatement of the reference's backward-search hot path (plain C11).
 * TEST INFRASTRUCTURE ONLY — see the header of oracle/fmoracle.h for scope and parity status.
 * Reference paths are relative to /root/reference/src/fmindex-collection/.
 */
#include "fmoracle.h"

#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define POPC(x) ((uint64_t)__builtin_popcountll(x))

static uint64_t round_up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }
static int bit_width_u64(uint64_t v) { return v ? 64 - __builtin_clzll(v) : 0; }
static uint64_t bit_ceil_u64(uint64_t v) { uint64_t r = 1; while (r < v) r <<= 1; return r; }

/* =====================================================================================
 * strings with rank support
 * ===================================================================================== */
enum family { FAM_IB, FAM_IBP, FAM_EPR, FAM_EPRV2, FAM_WAVELET, FAM_EPRH, FAM_FBV };

typedef struct ora_bitvector {               /* bitvector/Bitvector.h:30-179 */
    uint64_t* superblocks; uint64_t nsuper;  /* ones before bit 256k            */
    uint8_t*  blocks;      uint64_t nblocks; /* ones in [256*(j/4), 64j)        */
    uint64_t* bits;        uint64_t nbits;
    uint64_t  totalLength;
    uint64_t  cap;                           /* capacity in 64-bit blocks       */
} ora_bitvector;

struct ora_string {
    int layout, family, sigma, bitct;
    uint64_t n;
    /* blocked families */
    int bt;                  /* sizeof(block_t)                 */
    uint64_t K;              /* number of u64 words per block   */
    uint64_t bits_off;       /* offset of the u64 words         */
    uint64_t stride;         /* sizeof(Block)                   */
    uint64_t rows;           /* rows per block                  */
    uint64_t period;         /* rows per super-block            */
    uint8_t* blocks; uint64_t nblocks;
    uint64_t* super; uint64_t nsuper;          /* [nsuper][sigma] */
    /* EPR constants, string/InterleavedEPR.h:28-61 */
    uint64_t maskEven, bitMask; uint64_t* rb;
    /* wavelet, string/Wavelet.h:31-57 */
    uint64_t nnodes; ora_bitvector* node;
    /* EPRV3 / EPRV4 / EPRV5 / InterleavedEPRV7: bit-sliced 64-row blocks + counter levels in arrays of their own.
     * level L: element width lev_w[L] bytes, one [sigma] entry per 2^lev_shift[L] rows; super-block per 2^sshift rows.
     * V7 keeps level 0 inside the packed bits struct (hb_stride = 8*bitct + sigma, level0 at byte 8*bitct). */
    int nlev, lev_w[3], lev_shift[3], sshift, v3, v7;
    uint8_t* lev[3]; uint64_t lev_n[3];
    uint8_t* hbits; uint64_t hb_stride, hblocks;
    /* FlattenedBitvectors2L<sigma, l1_bits, 65536>: hbits = InBits per l1_bits rows (bitct bitsets of l1_bits bits, plane-major),
     * fl1 = u16 [nl1][sigma+1], super = u64 [nsuper][sigma+1]; entry [c] = number of symbols < c before the block */
    uint64_t l1_bits; uint16_t* fl1; uint64_t nl1;
};

static uint64_t blk_count(const ora_string* s, uint64_t b, uint64_t c) {
    const uint8_t* p = s->blocks + b * s->stride + c * (uint64_t)s->bt;
    switch (s->bt) {
    case 1: return *p;
    case 2: { uint16_t v; memcpy(&v, p, 2); return v; }
    default: { uint32_t v; memcpy(&v, p, 4); return v; }
    }
}
static void blk_set_count(ora_string* s, uint64_t b, uint64_t c, uint64_t v) {
    uint8_t* p = s->blocks + b * s->stride + c * (uint64_t)s->bt;
    switch (s->bt) {
    case 1: { uint8_t x = (uint8_t)v; memcpy(p, &x, 1); break; }
    case 2: { uint16_t x = (uint16_t)v; memcpy(p, &x, 2); break; }
    default: { uint32_t x = (uint32_t)v; memcpy(p, &x, 4); break; }
    }
}
static uint64_t blk_word(const ora_string* s, uint64_t b, uint64_t k) {
    uint64_t v; memcpy(&v, s->blocks + b * s->stride + s->bits_off + 8 * k, 8); return v;
}
static void blk_or_word(ora_string* s, uint64_t b, uint64_t k, uint64_t v) {
    uint8_t* p = s->blocks + b * s->stride + s->bits_off + 8 * k;
    uint64_t w; memcpy(&w, p, 8); w |= v; memcpy(p, &w, 8);
}
static uint64_t block_t_mask(int bt) { return bt == 4 ? 0xffffffffull : (bt == 2 ? 0xffffull : 0xffull); }

/* ---- bitvector::Bitvector (wavelet nodes) ------------------------------------------- */
static void bv_init(ora_bitvector* v, uint64_t cap_bits) {
    v->cap = cap_bits / 64 + 2;
    v->superblocks = calloc(v->cap / 4 + 2, 8);
    v->blocks = calloc(v->cap + 1, 1);
    v->bits = calloc(v->cap + 1, 8);
    v->nsuper = v->nblocks = v->nbits = 1;      /* Bitvector.h:31-33: arrays start with one zero entry */
    v->totalLength = 0;
}
static void bv_push_back(ora_bitvector* v, int bit) {   /* Bitvector.h:117-133 */
    if (bit) v->bits[v->nbits - 1] |= 1ull << (v->totalLength % 64);
    v->totalLength += 1;
    if (v->totalLength % 64 == 0) {
        uint64_t ct = (uint64_t)v->blocks[v->nblocks - 1] + POPC(v->bits[v->nbits - 1]);
        v->blocks[v->nblocks++] = (uint8_t)ct;
        v->bits[v->nbits++] = 0;
        if (v->totalLength % 256 == 0) {
            v->superblocks[v->nsuper] = v->superblocks[v->nsuper - 1] + ct;
            v->nsuper++;
            v->blocks[v->nblocks - 1] = 0;
        }
    }
}
static uint64_t bv_rank(const ora_bitvector* v, uint64_t idx) {   /* Bitvector.h:147-166 */
    uint64_t bitId = idx % 64, blockId = idx / 64, superblockId = blockId / 4;
    if (bitId == 0) return v->superblocks[superblockId] + v->blocks[blockId];
    return v->superblocks[superblockId] + v->blocks[blockId] + POPC(v->bits[blockId] << (64 - bitId));
}
static int bv_symbol(const ora_bitvector* v, uint64_t idx) {      /* Bitvector.h:139-145 */
    return (int)((v->bits[idx / 64] >> (idx % 64)) & 1);
}

/* ---- construction -------------------------------------------------------------------- */
static int layout_params(ora_string* s, int layout, int sigma) {
    s->layout = layout; s->sigma = sigma; s->bitct = bit_width_u64((uint64_t)sigma - 1);
    uint64_t align = 8;
    switch (layout) {
    case ORA_IB8:      s->family = FAM_IB;    s->bt = 1; break;
    case ORA_IB16:     s->family = FAM_IB;    s->bt = 2; break;
    case ORA_IB32:     s->family = FAM_IB;    s->bt = 4; break;
    case ORA_IB16A:    s->family = FAM_IB;    s->bt = 2; align = 64; break;
    case ORA_IBP16:    s->family = FAM_IBP;   s->bt = 2; break;
    case ORA_EPR8:     s->family = FAM_EPR;   s->bt = 1; break;
    case ORA_EPR16:    s->family = FAM_EPR;   s->bt = 2; break;
    case ORA_EPR32:    s->family = FAM_EPR;   s->bt = 4; break;
    case ORA_EPRV2_8:  s->family = FAM_EPRV2; s->bt = 1; break;
    case ORA_EPRV2_16: s->family = FAM_EPRV2; s->bt = 2; break;
    case ORA_EPRV2_32: s->family = FAM_EPRV2; s->bt = 4; break;
    case ORA_WAVELET:  s->family = FAM_WAVELET; s->bt = 0; return 0;
    case ORA_EPRV3_8: case ORA_EPRV3_16: case ORA_EPRV3_32: {          /* EPRV3.h:128-133: blocks_ per 64 rows, super-block per 2^(8*sizeof(block_t)) rows */
        s->family = FAM_EPRH; s->v3 = 1; s->bt = layout == ORA_EPRV3_8 ? 1 : (layout == ORA_EPRV3_16 ? 2 : 4);
        s->nlev = 1; s->lev_w[0] = s->bt; s->lev_shift[0] = 6; s->sshift = 8 * s->bt;
        s->hb_stride = 8 * (uint64_t)s->bitct; return 0;
    }
    case ORA_EPRV4:    /* EPRV4.h:27-42: u8 per 64 rows, u16 per 256, u32 per 65 536, super-block per 2^32 */
        s->family = FAM_EPRH; s->nlev = 3; s->lev_w[0] = 1; s->lev_w[1] = 2; s->lev_w[2] = 4;
        s->lev_shift[0] = 6; s->lev_shift[1] = 8; s->lev_shift[2] = 16; s->sshift = 32; s->hb_stride = 8 * (uint64_t)s->bitct; return 0;
    case ORA_EPRV5:    /* EPRV5.h:27-42: u8 per 64 rows, u16 per 256, super-block per 65 536 */
    case ORA_IEPRV7:   /* InterleavedEPRV7.h:23-26, :126-138: as V5 with the u8 level inside the packed bits struct */
        s->family = FAM_EPRH; s->nlev = 2; s->lev_w[0] = 1; s->lev_w[1] = 2;
        s->lev_shift[0] = 6; s->lev_shift[1] = 8; s->sshift = 16; s->v7 = layout == ORA_IEPRV7;
        s->hb_stride = 8 * (uint64_t)s->bitct + (s->v7 ? (uint64_t)sigma : 0); return 0;
    case ORA_FBV_64_64K: case ORA_FBV_512_64K: case ORA_FBV_2048_64K:   /* FlattenedBitvectors2L.h:23-36, :274-279 */
        s->family = FAM_FBV; s->l1_bits = layout == ORA_FBV_64_64K ? 64 : (layout == ORA_FBV_512_64K ? 512 : 2048);
        s->hb_stride = (uint64_t)s->bitct * s->l1_bits / 8; return 0;
    default: return -1;
    }
    uint64_t full = 1ull << (8 * s->bt);      /* 2^(8*sizeof(block_t)) */
    switch (s->family) {
    case FAM_IB: case FAM_IBP: s->K = (uint64_t)sigma; s->rows = 64; s->period = full; break;
    case FAM_EPRV2:            s->K = (uint64_t)s->bitct; s->rows = 64; s->period = full; break;
    case FAM_EPR:              s->K = 1; s->rows = 64 / (uint64_t)s->bitct;            /* InterleavedEPR.h:105 */
                               s->period = (full / s->rows) * s->rows; break;           /* InterleavedEPR.h:106 */
    }
    s->bits_off = round_up((uint64_t)sigma * (uint64_t)s->bt, 8);
    s->stride = round_up(s->bits_off + 8 * s->K, align);
    return 0;
}

static void build_ib(ora_string* s, const uint8_t* t, uint64_t n, int prefix) {
    /* string/InterleavedBitvector.h:64-94, string/InterleavedBitvectorPrefix.h:68-101 */
    uint64_t sigma = (uint64_t)s->sigma, m = block_t_mask(s->bt);
    s->nblocks = n / 64 + 1;
    s->nsuper = n / s->period + 1;
    s->blocks = calloc(s->nblocks, s->stride);
    s->super = calloc(s->nsuper * sigma, 8);
    uint64_t* sacc = calloc(sigma, 8);
    uint64_t* bacc = calloc(sigma, 8);
    uint64_t sb = 0;
    for (uint64_t size = 1; size <= n; ++size) {
        uint64_t blockId = size >> 6, bitId = size & 63;
        if (size % s->period == 0) {                 /* new super block + new (zero-count) block */
            ++sb;
            memcpy(s->super + sb * sigma, sacc, 8 * sigma);
            memset(bacc, 0, 8 * sigma);
        } else if (size % 64 == 0) {                 /* new block */
            for (uint64_t c = 0; c < sigma; ++c) blk_set_count(s, blockId, c, bacc[c] & m);
        }
        uint64_t symb = t[size - 1];
        for (uint64_t c = symb; c < (prefix ? sigma : symb + 1); ++c) {
            blk_or_word(s, blockId, c, 1ull << bitId);
            bacc[c] = (bacc[c] + 1) & m;
            sacc[c] += 1;
        }
    }
    free(sacc); free(bacc);
}

static void build_epr(ora_string* s, const uint8_t* t, uint64_t n) {
    /* string/InterleavedEPR.h:110-142 (always appends one trailing block + super-block row) */
    uint64_t sigma = (uint64_t)s->sigma, m = block_t_mask(s->bt), lf = s->rows, bc = (uint64_t)s->bitct;
    uint64_t cap_blocks = n / lf + 2, cap_super = n / s->period + 2;
    s->blocks = calloc(cap_blocks, s->stride);
    s->super = calloc(cap_super * sigma, 8);
    uint64_t* sacc = calloc(sigma, 8);
    uint64_t* bacc = calloc(sigma, 8);
    s->nblocks = 0; s->nsuper = 0;
    for (uint64_t size = 0; size < n;) {
        memcpy(s->super + (s->nsuper++) * sigma, sacc, 8 * sigma);
        memset(bacc, 0, 8 * sigma);
        for (uint64_t blockId = 0; blockId < s->period / lf && size < n; ++blockId) {
            uint64_t b = s->nblocks++;
            for (uint64_t c = 0; c < sigma; ++c) blk_set_count(s, b, c, bacc[c]);
            for (uint64_t bitId = 0; bitId < lf && size < n; ++bitId, ++size) {
                uint64_t symb = t[size];
                blk_or_word(s, b, 0, symb << (bc * bitId));
                bacc[symb] = (bacc[symb] + 1) & m;
                sacc[symb] += 1;
            }
        }
    }
    memcpy(s->super + (s->nsuper++) * sigma, sacc, 8 * sigma);
    { uint64_t b = s->nblocks++; for (uint64_t c = 0; c < sigma; ++c) blk_set_count(s, b, c, bacc[c]); }
    free(sacc); free(bacc);
    /* constants, InterleavedEPR.h:28-61 */
    uint64_t entries = 64 / bc;
    s->maskEven = 0; s->bitMask = 0;
    uint64_t chunkMaskEven = (1ull << bc) - 1, mask = 1ull << bc;
    for (uint64_t i = 0; i < entries; i += 2) {
        s->maskEven = (2 * bc >= 64 ? 0 : (s->maskEven << (2 * bc))) | chunkMaskEven;
        s->bitMask  = (2 * bc >= 64 ? 0 : (s->bitMask  << (2 * bc))) | mask;
    }
    s->rb = calloc(sigma, 8);
    for (uint64_t symb = 0; symb < sigma; ++symb) {
        uint64_t mk = symb | (1ull << bc), r = 0;
        for (uint64_t i = 0; i < entries; i += 2) r = (2 * bc >= 64 ? 0 : (r << (2 * bc))) | mk;
        s->rb[symb] = r;
    }
}

static void build_eprv2(ora_string* s, const uint8_t* t, uint64_t n) {
    /* string/InterleavedEPRV2.h:150-192 (trailing block only when n % 64 == 0) */
    uint64_t sigma = (uint64_t)s->sigma, m = block_t_mask(s->bt);
    uint64_t cap_blocks = n / 64 + 2, cap_super = n / s->period + 2;
    s->blocks = calloc(cap_blocks, s->stride);
    s->super = calloc(cap_super * sigma, 8);
    uint64_t* sacc = calloc(sigma, 8);
    uint64_t* bacc = calloc(sigma, 8);
    s->nblocks = 0; s->nsuper = 0;
    for (uint64_t size = 0; size < n;) {
        memcpy(s->super + (s->nsuper++) * sigma, sacc, 8 * sigma);
        memset(bacc, 0, 8 * sigma);
        for (uint64_t blockId = 0; blockId < s->period / 64 && size < n; ++blockId) {
            uint64_t b = s->nblocks++;
            for (uint64_t c = 0; c < sigma; ++c) blk_set_count(s, b, c, bacc[c]);
            for (uint64_t bitId = 0; bitId < 64 && size < n; ++bitId, ++size) {
                uint64_t symb = t[size];
                for (int i = 0; i < s->bitct; ++i) blk_or_word(s, b, (uint64_t)i, ((symb >> i) & 1) << bitId);
                bacc[symb] = (bacc[symb] + 1) & m;
                sacc[symb] += 1;
            }
        }
    }
    if (n % 64 == 0) {
        memcpy(s->super + (s->nsuper++) * sigma, sacc, 8 * sigma);
        uint64_t b = s->nblocks++;
        for (uint64_t c = 0; c < sigma; ++c) blk_set_count(s, b, c, bacc[c]);
    }
    free(sacc); free(bacc);
}

/* string/Wavelet.h:37-50: level b of symbol c -> (bit, node id) */
static void wavelet_step(const ora_string* s, uint64_t symb, int b, int* bit, uint64_t* id) {
    int bitId = s->bitct - b - 1;
    *bit = (int)((symb >> bitId) & 1);
    *id = ((1ull << b) - 1) + (symb >> (bitId + 1));
}

static void build_wavelet(ora_string* s, const uint8_t* t, uint64_t n) {
    /* string/Wavelet.h:60-71 */
    s->nnodes = bit_ceil_u64((uint64_t)s->sigma);
    s->node = calloc(s->nnodes, sizeof(ora_bitvector));
    /* capacity: node at level b holds at most n bits */
    for (uint64_t i = 0; i < s->nnodes; ++i) bv_init(&s->node[i], n);
    for (uint64_t p = 0; p < n; ++p) {
        for (int b = 0; b < s->bitct; ++b) {
            int bit; uint64_t id; wavelet_step(s, t[p], b, &bit, &id);
            bv_push_back(&s->node[id], bit);
        }
    }
}

/* ---- EPRV3/4/5/7 ---------------------------------------------------------------------- */
static uint8_t* eprh_level_ptr(const ora_string* s, int L, uint64_t entry) {
    if (L == 0 && s->v7) return s->hbits + entry * s->hb_stride + 8 * (uint64_t)s->bitct;
    return s->lev[L] + entry * (uint64_t)s->lev_w[L] * (uint64_t)s->sigma;
}
static uint64_t eprh_level_get(const ora_string* s, int L, uint64_t entry, uint64_t c) {
    const uint8_t* p = eprh_level_ptr(s, L, entry) + c * (uint64_t)s->lev_w[L];
    switch (s->lev_w[L]) {
    case 1: return *p;
    case 2: { uint16_t v; memcpy(&v, p, 2); return v; }
    default: { uint32_t v; memcpy(&v, p, 4); return v; }
    }
}
static void eprh_level_set(ora_string* s, int L, uint64_t entry, const uint64_t* acc) {
    for (uint64_t c = 0; c < (uint64_t)s->sigma; ++c) {
        uint8_t* p = eprh_level_ptr(s, L, entry) + c * (uint64_t)s->lev_w[L];
        switch (s->lev_w[L]) {                                   /* the accumulators have the level's own integer type */
        case 1: { uint8_t v = (uint8_t)acc[c]; *p = v; break; }
        case 2: { uint16_t v = (uint16_t)acc[c]; memcpy(p, &v, 2); break; }
        default: { uint32_t v = (uint32_t)acc[c]; memcpy(p, &v, 4); break; }
        }
    }
}
static uint64_t eprh_word(const ora_string* s, uint64_t b, int plane) {
    uint64_t v; memcpy(&v, s->hbits + b * s->hb_stride + 8 * (uint64_t)plane, 8); return v;
}
static void eprh_set_symbol(ora_string* s, uint64_t pos, uint64_t symb) {
    for (int i = 0; i < s->bitct; ++i) {
        uint8_t* p = s->hbits + (pos >> 6) * s->hb_stride + 8 * (uint64_t)i;
        uint64_t v; memcpy(&v, p, 8); v |= ((symb >> i) & 1ull) << (pos & 63); memcpy(p, &v, 8);
    }
}
static void build_eprh(ora_string* s, const uint8_t* t, uint64_t n) {
    uint64_t sigma = (uint64_t)s->sigma;
    uint64_t* sacc = calloc(sigma, 8);
    uint64_t* acc[3] = {calloc(sigma, 8), calloc(sigma, 8), calloc(sigma, 8)};
    if (s->v3) {
        /* EPRV3.h:151-186: loop over super-blocks / blocks, then one more super-block, block and InBits "for safety" —
         * the trailing block keeps the running count of the last super-block (not reset), which is what rank(n, c) reads
         * when n is a multiple of 64 */
        uint64_t per_super = (s->sshift >= 38 ? (1ull << 32) : (1ull << s->sshift) / 64);     /* blocks per super-block */
        s->hblocks = (n + 63) / 64 + 1;
        s->lev_n[0] = s->hblocks;
        s->nsuper = (n == 0 ? 0 : ((n - 1) >> s->sshift) + 1) + 1;
        s->hbits = calloc(s->hblocks * s->hb_stride + 8, 1);
        s->lev[0] = calloc(s->lev_n[0] * sigma * (uint64_t)s->lev_w[0] + 8, 1);
        s->super = calloc(s->nsuper * sigma, 8);
        uint64_t size = 0, nb = 0, nsb = 0;
        while (size < n) {
            memcpy(s->super + nsb++ * sigma, sacc, sigma * 8);
            memset(acc[0], 0, sigma * 8);
            for (uint64_t blockId = 0; blockId < per_super && size < n; ++blockId) {
                eprh_level_set(s, 0, nb++, acc[0]);
                for (uint64_t bitId = 0; bitId < 64 && size < n; ++bitId, ++size) {
                    eprh_set_symbol(s, size, t[size]);
                    acc[0][t[size]] += 1; sacc[t[size]] += 1;
                }
            }
        }
        memcpy(s->super + nsb++ * sigma, sacc, sigma * 8);
        eprh_level_set(s, 0, nb++, acc[0]);
    } else {
        /* EPRV4.h:56-99, EPRV5.h:49-97, InterleavedEPRV7.h:147-177: for size in [0, n]: open the blocks that start at `size`
         * (the highest level whose period divides `size` stores its accumulator, every level below it starts from zero) */
        s->hblocks = n / 64 + 1;
        for (int L = 0; L < s->nlev; ++L) s->lev_n[L] = (n >> s->lev_shift[L]) + 1;
        s->nsuper = (s->sshift >= 64 ? 0 : (n >> s->sshift)) + 1;
        s->hbits = calloc(s->hblocks * s->hb_stride + 8, 1);
        for (int L = (s->v7 ? 1 : 0); L < s->nlev; ++L) s->lev[L] = calloc(s->lev_n[L] * sigma * (uint64_t)s->lev_w[L] + 8, 1);
        s->super = calloc(s->nsuper * sigma, 8);
        uint64_t cnt[3] = {0, 0, 0}, nsb = 0;
        for (uint64_t size = 0; size <= n; ++size) {
            if ((size & ((1ull << s->sshift) - 1)) == 0) {
                memcpy(s->super + nsb++ * sigma, sacc, sigma * 8);
                for (int L = 0; L < s->nlev; ++L) { memset(acc[L], 0, sigma * 8); cnt[L]++; }       /* value-initialised entries */
            } else {
                for (int L = s->nlev - 1; L >= 0; --L) {
                    if ((size & ((1ull << s->lev_shift[L]) - 1)) != 0) continue;
                    eprh_level_set(s, L, cnt[L]++, acc[L]);
                    for (int l = 0; l < L; ++l) { memset(acc[l], 0, sigma * 8); cnt[l]++; }
                    break;
                }
            }
            if (size == n) continue;
            eprh_set_symbol(s, size, t[size]);
            for (int L = 0; L < s->nlev; ++L) acc[L][t[size]] += 1;
            sacc[t[size]] += 1;
        }
    }
    free(sacc); free(acc[0]); free(acc[1]); free(acc[2]);
}
/* EPRV3.h:55-68 symbol-match mask over the bit planes */
static uint64_t eprh_have(const ora_string* s, uint64_t b, uint64_t symb) {
    uint64_t r = ~0ull;
    for (int i = 0; i < s->bitct; ++i) r &= eprh_word(s, b, i) ^ (0 - ((~symb >> i) & 1));
    return r;
}
static uint64_t eprh_counters(const ora_string* s, uint64_t idx, uint64_t c) {
    uint64_t a = s->super[(s->sshift >= 64 ? 0 : idx >> s->sshift) * (uint64_t)s->sigma + c];
    for (int L = 0; L < s->nlev; ++L) a += eprh_level_get(s, L, idx >> s->lev_shift[L], c);
    return a;
}

/* ---- FlattenedBitvectors2L ---------------------------------------------------------------- */
static uint64_t fbv_word(const ora_string* s, uint64_t blk, int plane, uint64_t w) {
    uint64_t v; memcpy(&v, s->hbits + blk * s->hb_stride + (uint64_t)plane * (s->l1_bits / 8) + 8 * w, 8); return v;
}
static uint64_t fbv_have(const ora_string* s, uint64_t blk, uint64_t w, uint64_t symb) {      /* ternarylogic.h:731-744 mark_exact_large */
    uint64_t r = ~0ull;
    for (int i = 0; i < s->bitct; ++i) r &= fbv_word(s, blk, i, w) ^ (0 - ((~symb >> i) & 1));
    return r;
}
static void build_fbv(ora_string* s, const uint8_t* t, uint64_t n) {      /* FlattenedBitvectors2L.h:131-192 */
    const uint64_t sig1 = (uint64_t)s->sigma + 1, per = 65536 / s->l1_bits;
    s->nsuper = n / 65536 + 1;
    s->nl1 = s->hblocks = s->nsuper * per;
    s->hbits = calloc(s->hblocks * s->hb_stride + 8, 1);
    s->fl1 = calloc(s->nl1 * sig1, 2);
    s->super = calloc(s->nsuper * sig1, 8);
    for (uint64_t i = 0; i < n; ++i)                                        /* InBits::setSymbol, :76-82 */
        for (int j = 0; j < s->bitct; ++j) {
            uint8_t* p = s->hbits + (i / s->l1_bits) * s->hb_stride + (uint64_t)j * (s->l1_bits / 8) + ((i % s->l1_bits) / 64) * 8;
            uint64_t v; memcpy(&v, p, 8); v |= (uint64_t)((t[i] >> j) & 1) << (i & 63); memcpy(p, &v, 8);
        }
    uint64_t* l0acc = calloc(sig1, 8); uint64_t* acc = calloc(sig1, 8);
    for (uint64_t sb = 0; sb < s->nsuper; ++sb) {
        memcpy(s->super + sb * sig1, l0acc, sig1 * 8);
        memset(acc, 0, sig1 * 8);
        for (uint64_t i = 0; i < per; ++i) {
            uint64_t blk = sb * per + i;
            for (uint64_t c = 0; c < sig1; ++c) s->fl1[blk * sig1 + c] = (uint16_t)acc[c];
            uint64_t a = 0;                                                 /* all_ranks(l1_bits): the unused tail of the last block counts as symbol 0 */
            for (uint64_t c = 0; c + 1 < sig1; ++c) {
                uint64_t cnt = 0;
                for (uint64_t w = 0; w < s->l1_bits / 64; ++w) cnt += POPC(fbv_have(s, blk, w, c));
                a += cnt; acc[c + 1] += a;
            }
        }
        for (uint64_t c = 0; c < sig1; ++c) l0acc[c] += acc[c];
    }
    free(l0acc); free(acc);
}
/* positions < bit of block `blk` holding a symbol for which pred: exact == 1: symbol == symb, else symbol < symb */
static uint64_t fbv_inblock(const ora_string* s, uint64_t blk, uint64_t bit, uint64_t symb, int exact) {
    uint64_t r = 0;
    for (uint64_t w = 0; w * 64 < bit; ++w) {
        uint64_t m = 0;
        if (exact) m = fbv_have(s, blk, w, symb);
        else for (uint64_t d = 0; d < symb; ++d) m |= fbv_have(s, blk, w, d);
        uint64_t k = bit - w * 64;
        r += POPC(k >= 64 ? m : (m & ((1ull << k) - 1)));
    }
    return r;
}

ora_string* ora_string_build(int layout, int sigma, const uint8_t* symbols, uint64_t n) {
    if (sigma < 2 || sigma > 256) return NULL;
    for (uint64_t i = 0; i < n; ++i) if (symbols[i] >= sigma) return NULL;
    ora_string* s = calloc(1, sizeof *s);
    if (layout_params(s, layout, sigma) != 0) { free(s); return NULL; }
    s->n = n;
    switch (s->family) {
    case FAM_IB:      build_ib(s, symbols, n, 0); break;
    case FAM_IBP:     build_ib(s, symbols, n, 1); break;
    case FAM_EPR:     build_epr(s, symbols, n); break;
    case FAM_EPRV2:   build_eprv2(s, symbols, n); break;
    case FAM_WAVELET: build_wavelet(s, symbols, n); break;
    case FAM_EPRH:    build_eprh(s, symbols, n); break;
    case FAM_FBV:     build_fbv(s, symbols, n); break;
    }
    return s;
}

void ora_string_free(ora_string* s) {
    if (!s) return;
    free(s->blocks); free(s->super); free(s->rb);
    free(s->hbits); free(s->lev[0]); free(s->lev[1]); free(s->lev[2]); free(s->fl1);
    if (s->node) {
        for (uint64_t i = 0; i < s->nnodes; ++i) { free(s->node[i].superblocks); free(s->node[i].blocks); free(s->node[i].bits); }
        free(s->node);
    }
    free(s);
}
/* Re-homes the big arrays of a string so that their pages are spread over the NUMA nodes of the threads that will search them: a new
 * buffer is first touched by `nthreads` OpenMP threads in a static schedule (the construction above runs on one thread, so everything it
 * allocated sits on that thread's node and every other socket's queries cross the interconnect).  Contents are unchanged.  Test
 * infrastructure for bench.py's cpu_baseline leg — the reference itself does nothing of the kind. */
static void* spread_copy(void* old, uint64_t bytes, int nthreads) {
    if (!old || bytes < (1u << 20)) return old;
    uint8_t* fresh = malloc(bytes);
    if (!fresh) return old;
    const uint64_t page = 1u << 21;                      /* a transparent huge page */
    const int64_t npages = (int64_t)((bytes + page - 1) / page);
    #pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
    for (int64_t p = 0; p < npages; ++p) {
        const uint64_t o = (uint64_t)p * page, len = o + page <= bytes ? page : bytes - o;
        memcpy(fresh + o, (const uint8_t*)old + o, len);
    }
    free(old);
    return fresh;
}
void ora_index_spread(ora_index* x, int nthreads) {
    ora_string* two[2] = {x->bwt, x->bwt_rev};
    for (int k = 0; k < 2; ++k) {
        ora_string* s = two[k];
        if (!s) continue;
        if (s->blocks) s->blocks = spread_copy(s->blocks, (s->family == FAM_IB || s->family == FAM_IBP ? s->nblocks : (s->n / (s->rows ? s->rows : 64) + 2)) * s->stride, nthreads);
        if (s->node) for (uint64_t i = 0; i < s->nnodes; ++i) {
            ora_bitvector* v = &s->node[i];
            v->superblocks = spread_copy(v->superblocks, (v->cap / 4 + 2) * 8, nthreads);
            v->blocks = spread_copy(v->blocks, v->cap + 1, nthreads);
            v->bits = spread_copy(v->bits, (v->cap + 1) * 8, nthreads);
        }
    }
}
uint64_t ora_string_size(const ora_string* s) { return s->n; }
int ora_string_sigma(const ora_string* s) { return s->sigma; }
int ora_string_layout(const ora_string* s) { return s->layout; }
uint64_t ora_string_block_stride(const ora_string* s) { return s->stride; }
uint64_t ora_string_bits_offset(const ora_string* s) { return s->bits_off; }

int ora_string_raw(const ora_string* s, int part, const void** ptr, uint64_t* bytes) {
    if (s->family == FAM_WAVELET) {
        uint64_t node = (uint64_t)part / 4; int what = part % 4;
        if (node >= s->nnodes) return -1;
        const ora_bitvector* v = &s->node[node];
        switch (what) {
        case 0: *ptr = v->superblocks; *bytes = v->nsuper * 8; return 0;
        case 1: *ptr = v->blocks;      *bytes = v->nblocks; return 0;
        case 2: *ptr = v->bits;        *bytes = v->nbits * 8; return 0;
        default: *ptr = &v->totalLength; *bytes = 8; return 0;
        }
    }
    if (s->family == FAM_FBV) {
        if (part == 0) { *ptr = s->hbits; *bytes = s->hblocks * s->hb_stride; return 0; }
        if (part == 1) { *ptr = s->super; *bytes = s->nsuper * ((uint64_t)s->sigma + 1) * 8; return 0; }
        if (part == 2) { *ptr = s->fl1; *bytes = s->nl1 * ((uint64_t)s->sigma + 1) * 2; return 0; }
        return -1;
    }
    if (s->family == FAM_EPRH) {
        if (part == 0) { *ptr = s->hbits; *bytes = s->hblocks * s->hb_stride; return 0; }
        if (part == 1) { *ptr = s->super; *bytes = s->nsuper * (uint64_t)s->sigma * 8; return 0; }
        if (part >= 2 && part < 2 + s->nlev) {
            int L = part - 2;
            *ptr = s->lev[L]; *bytes = s->lev[L] ? s->lev_n[L] * (uint64_t)s->sigma * (uint64_t)s->lev_w[L] : 0; return 0;
        }
        return -1;
    }
    if (part == 0) { *ptr = s->blocks; *bytes = s->nblocks * s->stride; return 0; }
    if (part == 1) { *ptr = s->super; *bytes = s->nsuper * (uint64_t)s->sigma * 8; return 0; }
    return -1;
}

/* ---- in-block kernels ---------------------------------------------------------------- */
/* EPRV2 symbol-match mask, string/InterleavedEPRV2.h:28-46 */
static uint64_t eprv2_have(const ora_string* s, uint64_t b, uint64_t symb) {
    uint64_t r = ~0ull;
    for (int i = 0; i < s->bitct; ++i) {
        uint64_t inv = (~symb >> i) & 1;
        r &= blk_word(s, b, (uint64_t)i) ^ (0 - inv);
    }
    return r;
}
/* bitset<64> << k for k in [0,64] */
static uint64_t shl64(uint64_t v, uint64_t k) { return k >= 64 ? 0 : v << k; }

/* EPR in-block prefix rank, string/InterleavedEPR.h:72-89 */
static uint64_t epr_block_prefix(const ora_string* s, uint64_t b, uint64_t idx, uint64_t symb) {
    if (symb == 0) return 0;
    symb -= 1;
    uint64_t bc = (uint64_t)s->bitct, in = blk_word(s, b, 0);
    uint64_t te = ((s->rb[symb] - (in & s->maskEven)) & s->bitMask) >> bc;
    uint64_t to = (s->rb[symb] - ((in >> bc) & s->maskEven)) & s->bitMask;
    uint64_t epr = (te | to) & ((1ull << (idx * bc)) - 1);
    uint64_t ct = POPC(epr);
    for (uint64_t i = 0; i <= symb; ++i) ct += blk_count(s, b, i);
    return ct;
}

uint64_t ora_rank(const ora_string* s, uint64_t idx, uint64_t symb) {
    uint64_t sigma = (uint64_t)s->sigma;
    switch (s->family) {
    case FAM_IB: {      /* InterleavedBitvector.h:112-117, :23-26 */
        uint64_t b = idx >> 6, sb = idx / s->period, bit = idx & 63;
        return s->super[sb * sigma + symb] + blk_count(s, b, symb) + POPC(blk_word(s, b, symb) << (63 - bit));
    }
    case FAM_IBP: {     /* InterleavedBitvectorPrefix.h:126-135, :27-36 */
        uint64_t b = idx >> 6, sb = idx / s->period, bit = idx & 63;
        uint64_t w = blk_word(s, b, symb), blk = blk_count(s, b, symb);
        if (symb > 0) { w &= ~blk_word(s, b, symb - 1); blk = (blk - blk_count(s, b, symb - 1)) & block_t_mask(s->bt); }
        uint64_t r = blk + POPC(w << (63 - bit)) + s->super[sb * sigma + symb];
        if (symb > 0) r -= s->super[sb * sigma + symb - 1];
        return r;
    }
    case FAM_EPR: {     /* InterleavedEPR.h:160-167 */
        uint64_t b = idx / s->rows, sb = idx / s->period, bit = idx % s->rows;
        return epr_block_prefix(s, b, bit, symb + 1) - epr_block_prefix(s, b, bit, symb) + s->super[sb * sigma + symb];
    }
    case FAM_EPRV2: {   /* InterleavedEPRV2.h:197-202, :69-74 */
        uint64_t b = idx >> 6, sb = idx / s->period, bit = idx & 63;
        return blk_count(s, b, symb) + POPC(shl64(eprv2_have(s, b, symb), 64 - bit)) + s->super[sb * sigma + symb];
    }
    case FAM_FBV: {     /* FlattenedBitvectors2L.h:209-224 */
        uint64_t blk = idx / s->l1_bits, sb = idx / 65536, sig1 = sigma + 1;
        return s->super[sb * sig1 + symb + 1] + s->fl1[blk * sig1 + symb + 1] + fbv_inblock(s, blk, idx % s->l1_bits, symb, 1)
             - s->super[sb * sig1 + symb] - s->fl1[blk * sig1 + symb];
    }
    case FAM_EPRH:      /* EPRV3.h:205-213, EPRV4.h:128-142, EPRV5.h:126-139, InterleavedEPRV7.h:190-199 */
        return eprh_counters(s, idx, symb) + POPC(shl64(eprh_have(s, idx >> 6, symb), 64 - (idx & 63)));
    default: {          /* Wavelet.h:104-119 */
        for (int b = 0; b < s->bitct; ++b) {
            int bit; uint64_t id; wavelet_step(s, symb, b, &bit, &id);
            uint64_t r = bv_rank(&s->node[id], idx);
            idx = bit ? r : idx - r;
        }
        return idx;
    }
    }
}

uint64_t ora_prefix_rank(const ora_string* s, uint64_t idx, uint64_t symb) {
    uint64_t sigma = (uint64_t)s->sigma;
    switch (s->family) {
    case FAM_IB: {      /* InterleavedBitvector.h:119-128, :28-37 */
        uint64_t b = idx >> 6, sb = idx / s->period, bit = idx & 63, w = 0, a = 0;
        for (uint64_t i = 0; i < symb; ++i) { w |= blk_word(s, b, i); a += blk_count(s, b, i) + s->super[sb * sigma + i]; }
        return a + POPC(w << (63 - bit));
    }
    case FAM_IBP: {     /* InterleavedBitvectorPrefix.h:144-151 */
        if (symb == 0) return 0;
        symb -= 1;
        uint64_t b = idx >> 6, sb = idx / s->period, bit = idx & 63;
        return blk_count(s, b, symb) + POPC(blk_word(s, b, symb) << (63 - bit)) + s->super[sb * sigma + symb];
    }
    case FAM_EPR: {     /* InterleavedEPR.h:169-178 */
        uint64_t b = idx / s->rows, sb = idx / s->period, bit = idx % s->rows, a = 0;
        for (uint64_t i = 0; i < symb; ++i) a += s->super[sb * sigma + i];
        return epr_block_prefix(s, b, bit, symb) + a;
    }
    case FAM_EPRV2: {   /* InterleavedEPRV2.h:204-213, :76-86 */
        uint64_t b = idx >> 6, sb = idx / s->period, bit = idx & 63, w = 0, a = 0;
        for (uint64_t i = 0; i < symb; ++i) { w |= eprv2_have(s, b, i); a += blk_count(s, b, i) + s->super[sb * sigma + i]; }
        return a + POPC(shl64(w, 64 - bit));
    }
    case FAM_FBV: {     /* FlattenedBitvectors2L.h:226-239 */
        uint64_t blk = idx / s->l1_bits, sb = idx / 65536, sig1 = sigma + 1;
        return fbv_inblock(s, blk, idx % s->l1_bits, symb, 0) + s->super[sb * sig1 + symb] + s->fl1[blk * sig1 + symb];
    }
    case FAM_EPRH: {    /* EPRV3.h:215-227, EPRV4.h:144-160, EPRV5.h:141-157, InterleavedEPRV7.h:201-214 */
        uint64_t w = 0, a = 0;
        for (uint64_t i = 0; i < symb; ++i) { w |= eprh_have(s, idx >> 6, i); a += eprh_counters(s, idx, i); }
        return a + POPC(shl64(w, 64 - (idx & 63)));
    }
    default: {          /* Wavelet.h:121-141 */
        if (symb == 0) return 0;
        symb -= 1;
        uint64_t a = 0;
        for (int b = 0; b < s->bitct; ++b) {
            int bit; uint64_t id; wavelet_step(s, symb, b, &bit, &id);
            uint64_t r = bv_rank(&s->node[id], idx);
            if (bit == 0) idx = idx - r; else { a += idx - r; idx = r; }
        }
        return a + idx;
    }
    }
}

uint64_t ora_symbol(const ora_string* s, uint64_t idx) {
    switch (s->family) {
    case FAM_IB: case FAM_IBP: {   /* InterleavedBitvector.h:105-110, :39-47 */
        uint64_t i = idx + 1, b = i >> 6, bit = 1ull << (i & 63);
        for (uint64_t c = 0; c + 1 < (uint64_t)s->sigma; ++c) if (blk_word(s, b, c) & bit) return c;
        return (uint64_t)s->sigma - 1;
    }
    case FAM_EPR: {                /* InterleavedEPR.h:154-158, :91-97 */
        uint64_t b = idx / s->rows, bit = idx % s->rows, bc = (uint64_t)s->bitct;
        return (blk_word(s, b, 0) >> (bit * bc)) & ((1ull << bc) - 1);
    }
    case FAM_EPRV2: {              /* InterleavedEPRV2.h:191-195, :98-105 */
        uint64_t b = idx >> 6, bit = idx & 63, symb = 0;
        for (int i = s->bitct; i > 0; --i) symb = (symb << 1) | ((blk_word(s, b, (uint64_t)i - 1) >> bit) & 1);
        return symb;
    }
    case FAM_FBV: {                /* FlattenedBitvectors2L.h:197-206, :41-50 */
        uint64_t symb = 0, blk = idx / s->l1_bits, bit = idx % s->l1_bits;
        for (int i = s->bitct; i > 0; --i) symb = (symb << 1) | ((fbv_word(s, blk, i - 1, bit / 64) >> (bit & 63)) & 1);
        return symb;
    }
    case FAM_EPRH: {               /* EPRV3.h:199-203, :45-52 */
        uint64_t symb = 0;
        for (int i = s->bitct; i > 0; --i) symb = (symb << 1) | ((eprh_word(s, idx >> 6, i - 1) >> (idx & 63)) & 1);
        return symb;
    }
    default: {                     /* Wavelet.h:77-102 */
        uint64_t symb = 0;
        for (int b = 0; b < s->bitct; ++b) {
            uint64_t id = ((1ull << b) - 1) + symb;
            int bit = bv_symbol(&s->node[id], idx);
            uint64_t r = bv_rank(&s->node[id], idx);
            symb = (symb << 1) | (uint64_t)bit;
            idx = bit ? r : idx - r;
        }
        return symb;
    }
    }
}

/* string/concepts.h:50-64 — the mathematical contract; several reference variants get prs wrong */
void ora_all_ranks_and_prefix_ranks(const ora_string* s, uint64_t idx, uint64_t* rs, uint64_t* prs) {
    uint64_t acc = 0;
    for (uint64_t c = 0; c < (uint64_t)s->sigma; ++c) {
        rs[c] = ora_rank(s, idx, c);
        prs[c] = acc;
        acc += rs[c];
    }
}

/* =====================================================================================
 * sampled suffix array: SparseArray<tuple<u32,u32>, Bitvector2L<512,65536>>
 * ===================================================================================== */
static void dense_init(ora_dense_vector* v, uint64_t largest, uint64_t divisor, uint64_t count) {
    /* DenseVector.h:57-61 */
    v->largestValue = largest; v->commonDivisor = divisor;
    v->bits = (uint8_t)bit_width_u64(largest / divisor);
    v->bitCount = 0;
    v->nwords = 0;
    v->data = calloc((uint64_t)v->bits * count / 64 + 2, 8);
}
static void dense_push(ora_dense_vector* v, uint64_t value) {   /* DenseVector.h:124-144 */
    value /= v->commonDivisor;
    uint64_t empty = v->nwords * 64 - v->bitCount;
    if (empty == 0) { v->data[v->nwords++] = value; v->bitCount += v->bits; return; }
    if (empty >= v->bits) { v->data[v->nwords - 1] |= value << (64 - empty); v->bitCount += v->bits; return; }
    v->data[v->nwords - 1] |= value << (64 - empty);
    v->data[v->nwords++] = value >> empty;
    v->bitCount += v->bits;
}
uint64_t ora_dense_access(const ora_dense_vector* v, uint64_t i) {   /* DenseVector.h:154-182 */
    uint64_t begin = i * v->bits, end = begin + v->bits - 1;
    uint64_t startI = begin / 64, endI = end / 64, off = begin % 64;
    uint64_t width = end - begin + 1;
    uint64_t mask = width >= 64 ? ~0ull : (1ull << width) - 1;
    uint64_t value;
    if (startI == endI) value = (v->data[startI] >> off) & mask;
    else value = ((v->data[startI] >> off) | (v->data[endI] << (64 - off))) & mask;
    return value * v->commonDivisor;
}
static uint64_t gcd_u64(uint64_t a, uint64_t b) { while (b) { uint64_t t = a % b; a = b; b = t; } return a; }
/* DenseVector on its own (the reference tests it apart from the SparseArray, checkDenseVector.cpp:8-82):
 * largest == 0 and divisor == 0: DenseVector{span} (DenseVector.h:84-99: largest value and gcd of the values); else DenseVector(largest, divisor) + push_back (:57-61) */
ora_dense_vector* ora_dense_build(const uint64_t* values, uint64_t n, uint64_t largest, uint64_t divisor) {
    ora_dense_vector* v = calloc(1, sizeof *v);
    if (largest == 0 && divisor == 0) {
        for (uint64_t i = 0; i < n; ++i) { if (values[i] > largest) largest = values[i]; divisor = gcd_u64(divisor, values[i]); }
        if (divisor == 0) divisor = 1;
    }
    dense_init(v, largest, divisor, n);
    for (uint64_t i = 0; i < n; ++i) dense_push(v, values[i]);
    return v;
}
uint64_t ora_dense_size(const ora_dense_vector* v) { return v->bits ? v->bitCount / v->bits : 0; }   /* DenseVector.h: size() */
ora_dense_vector* ora_dense_concat(const ora_dense_vector* a, const ora_dense_vector* b) {           /* DenseVector.h:38-50 */
    ora_dense_vector* v = calloc(1, sizeof *v);
    const uint64_t na = ora_dense_size(a), nb = ora_dense_size(b);
    dense_init(v, a->largestValue > b->largestValue ? a->largestValue : b->largestValue, gcd_u64(a->commonDivisor, b->commonDivisor), na + nb);
    for (uint64_t i = 0; i < na; ++i) dense_push(v, ora_dense_access(a, i));
    for (uint64_t i = 0; i < nb; ++i) dense_push(v, ora_dense_access(b, i));
    return v;
}
void ora_dense_free(ora_dense_vector* v) { if (v) { free(v->data); free(v); } }

ora_sparse* ora_sparse_build(uint64_t n, const uint8_t* has, const uint64_t* seq, const uint64_t* pos) {
    ora_sparse* s = calloc(1, sizeof *s);
    s->n = n;
    /* Bitvector2L<512,65536>, bitvector/Bitvector2L.h:37-81 */
    s->nl0 = n / 65536 + 1; s->nl1 = n / 512 + 1; s->nbitwords = (n / 512 + 1) * 8;
    s->l0 = calloc(s->nl0 + 1, 8); s->l1 = calloc(s->nl1 + 1, 2); s->bits = calloc(s->nbitwords, 8);
    for (uint64_t i = 0; i < n; ++i) if (has[i]) s->bits[i / 64] |= 1ull << (i % 64);
    uint64_t l1a = 0, total = 0, nblk = (n + 511) / 512;
    for (uint64_t b = 0; b < nblk; ++b) {
        uint64_t l0_id = total / 65536;
        uint64_t cnt = 0; for (int w = 0; w < 8; ++w) cnt += POPC(s->bits[b * 8 + (uint64_t)w]);
        total += 512; l1a += cnt;
        if (b + 1 < s->nl1 + 1) s->l1[b + 1] = (uint16_t)l1a;
        if (total % 65536 == 0) {
            if (l0_id + 1 < s->nl0 + 1) s->l0[l0_id + 1] = s->l0[l0_id] + l1a;
            s->l1[b + 1] = 0; l1a = 0;
        }
    }
    /* DenseMultiVector ctor, DenseMultiVector.h:65-103 */
    uint64_t largest[2] = {0, 0}, divisor[2] = {0, 0}, ct = 0;
    for (uint64_t i = 0; i < n; ++i) if (has[i]) {
        if (seq[i] > largest[0]) largest[0] = seq[i];
        if (pos[i] > largest[1]) largest[1] = pos[i];
        divisor[0] = gcd_u64(divisor[0], seq[i]); divisor[1] = gcd_u64(divisor[1], pos[i]);
        ++ct;
    }
    for (int f = 0; f < 2; ++f) { if (divisor[f] == 0) divisor[f] = 1; if (largest[f] == 0) largest[f] = 1; dense_init(&s->field[f], largest[f], divisor[f], ct); }
    for (uint64_t i = 0; i < n; ++i) if (has[i]) { dense_push(&s->field[0], seq[i]); dense_push(&s->field[1], pos[i]); }
    s->nvalues = ct;
    return s;
}
void ora_sparse_free(ora_sparse* s) {
    if (!s) return;
    free(s->l0); free(s->l1); free(s->bits); free(s->field[0].data); free(s->field[1].data); free(s);
}
static uint64_t sparse_rank(const ora_sparse* s, uint64_t idx) {    /* Bitvector2L.h:123-142 */
    uint64_t bitId = idx % 512, l1Id = idx / 512, l0Id = idx / 65536, cnt = 0;
    const uint64_t* w = s->bits + l1Id * 8;
    for (uint64_t k = 0; k < bitId / 64; ++k) cnt += POPC(w[k]);
    if (bitId % 64) cnt += POPC(w[bitId / 64] << (64 - bitId % 64));
    return s->l0[l0Id] + s->l1[l1Id] + cnt;
}
uint64_t ora_sparse_rank(const ora_sparse* s, uint64_t idx) { return sparse_rank(s, idx); }       /* (the presence bitvector's rank, for bitvector/unittest.cpp's vectors) */
int ora_sparse_value(const ora_sparse* s, uint64_t idx, uint64_t* seq, uint64_t* pos) {   /* SparseArray.h:63-70 */
    if (!((s->bits[idx / 64] >> (idx % 64)) & 1)) return 0;
    uint64_t r = sparse_rank(s, idx);
    *seq = ora_dense_access(&s->field[0], r);
    *pos = ora_dense_access(&s->field[1], r);
    return 1;
}

/* =====================================================================================
 * suffix array (stands in for libsais: utils.h:97-129 — any correct suffix sorter gives
 * the same SA; order = plain byte string order, a proper prefix sorts first)
 * ===================================================================================== */
/* Prefix doubling, refining only the groups that are still tied (Larsson & Sadakane's scheme with plain sorts): the first round orders all
 * suffixes by their first 7 symbols (LSD radix sort on the packed key), every later round sorts each tied group by the rank of the suffix h
 * positions on.  A long run of one symbol costs log2(run) rounds over ITS rows only. */
typedef struct { uint64_t key; uint32_t idx; } sa_item;
typedef struct { uint32_t key, idx; } sa_pair;
typedef struct { uint32_t s, e; } sa_group;
static int sa_pair_cmp(const void* a, const void* b) {
    const sa_pair* x = a; const sa_pair* y = b;
    return x->key < y->key ? -1 : (x->key > y->key ? 1 : 0);
}
static void sa_radix_sort(sa_item* it, sa_item* tmp, uint64_t n) {            /* stable, 16 bits per pass, keys below 2^63 */
    uint64_t* count = malloc(65537 * 8);
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 16 * pass;
        memset(count, 0, 65537 * 8);
        for (uint64_t i = 0; i < n; ++i) count[((it[i].key >> shift) & 0xffff) + 1]++;
        if (count[1] == n) continue;                                          /* every key has a zero digit here */
        for (int d = 0; d < 65536; ++d) count[d + 1] += count[d];
        for (uint64_t i = 0; i < n; ++i) tmp[count[(it[i].key >> shift) & 0xffff]++] = it[i];
        memcpy(it, tmp, n * sizeof *it);
    }
    free(count);
}
int ora_suffix_array(const uint8_t* text, uint64_t n, uint64_t* sa) {
    if (n == 0) return 0;
    if (n >= (1ull << 31)) return -1;
    sa_item* it = malloc(n * sizeof *it);
    sa_item* tmp = malloc(n * sizeof *tmp);
    uint32_t* rank = malloc(n * 4);
    uint32_t* idx = malloc(n * 4);
    /* initial key: first 7 symbols, 9 bits each (symbol+1, 0 = past the end: a proper prefix sorts first) */
    uint64_t h = 7;
    #pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        uint64_t k = 0;
        for (uint64_t j = 0; j < h; ++j) k = (k << 9) | ((uint64_t)i + j < n ? (uint64_t)text[i + j] + 1 : 0);
        it[i].key = k; it[i].idx = (uint32_t)i;
    }
    sa_radix_sort(it, tmp, n);
    free(tmp);
    /* rank = index of the first item with an equal key; tied groups are collected */
    uint64_t ngroups = 0, cap = 1024;
    sa_group* groups = malloc(cap * sizeof *groups);
    uint64_t start = 0;
    for (uint64_t i = 0; i <= n; ++i) {
        if (i == n || (i > 0 && it[i].key != it[i - 1].key)) {
            if (i - start > 1) {
                if (ngroups == cap) { cap *= 2; groups = realloc(groups, cap * sizeof *groups); }
                groups[ngroups].s = (uint32_t)start; groups[ngroups].e = (uint32_t)i; ++ngroups;
            }
            start = i;
        }
        if (i < n) { rank[it[i].idx] = (uint32_t)start; idx[i] = it[i].idx; }
    }
    free(it);
    sa_pair* pr = malloc(n * sizeof *pr);
    sa_group* next = malloc(cap * sizeof *next);
    uint64_t next_cap = cap;
    while (ngroups) {
        /* keys from the ranks of the previous round (no rank is written before every key of this round is read) */
        #pragma omp parallel for schedule(dynamic, 64)
        for (int64_t g = 0; g < (int64_t)ngroups; ++g)
            for (uint32_t i = groups[g].s; i < groups[g].e; ++i) {
                const uint64_t p = idx[i];
                pr[i].key = p + h < n ? rank[p + h] + 1u : 0u;
                pr[i].idx = (uint32_t)p;
            }
        #pragma omp parallel for schedule(dynamic, 16)
        for (int64_t g = 0; g < (int64_t)ngroups; ++g) qsort(pr + groups[g].s, groups[g].e - groups[g].s, sizeof *pr, sa_pair_cmp);
        uint64_t nn = 0;
        for (uint64_t g = 0; g < ngroups; ++g) {
            uint32_t st = groups[g].s;
            for (uint32_t i = groups[g].s; i <= groups[g].e; ++i) {
                if (i == groups[g].e || (i > groups[g].s && pr[i].key != pr[i - 1].key)) {
                    if (i - st > 1) {
                        if (nn == next_cap) { next_cap *= 2; next = realloc(next, next_cap * sizeof *next); }
                        next[nn].s = st; next[nn].e = i; ++nn;
                    }
                    st = i;
                }
                if (i < groups[g].e) { idx[i] = pr[i].idx; rank[pr[i].idx] = st; }
            }
        }
        sa_group* sw = groups; groups = next; next = sw;
        uint64_t sc = cap; cap = next_cap; next_cap = sc;
        ngroups = nn; h *= 2;
    }
    for (uint64_t i = 0; i < n; ++i) sa[i] = idx[i];
    free(pr); free(rank); free(idx); free(groups); free(next);
    return 0;
}
void ora_bwt_from_sa(const uint8_t* text, uint64_t n, const uint64_t* sa, uint8_t* bwt) {   /* utils.h:145-163 */
    for (uint64_t i = 0; i < n; ++i) bwt[i] = text[(sa[i] + n - 1) % n];
}

/* =====================================================================================
 * FMIndex / BiFMIndex
 * ===================================================================================== */
static void compute_C(ora_index* x) {   /* utils.h:199-206 */
    for (int c = 0; c <= x->sigma; ++c) x->C[c] = ora_prefix_rank(x->bwt, x->n, (uint64_t)c);
}

ora_index* ora_index_from_bwt(int layout, int sigma, const uint8_t* bwt, const uint8_t* bwt_rev, uint64_t n,
                              const uint8_t* has, const uint64_t* seq, const uint64_t* pos) {
    ora_index* x = calloc(1, sizeof *x);
    x->sigma = sigma; x->layout = layout; x->n = n; x->bidirectional = bwt_rev != NULL;
    x->bwt = ora_string_build(layout, sigma, bwt, n);
    if (!x->bwt) { free(x); return NULL; }
    if (bwt_rev) { x->bwt_rev = ora_string_build(layout, sigma, bwt_rev, n); if (!x->bwt_rev) { ora_index_free(x); return NULL; } }
    compute_C(x);
    x->sa = has ? ora_sparse_build(n, has, seq, pos) : NULL;
    return x;
}

ora_index* ora_index_build(int layout, int sigma, const uint8_t* seqs, const uint64_t* seq_off, uint64_t nseq,
                           uint64_t sampling_rate, int bidirectional) {
    /* createSequences: utils.h:382-411 / :413-464 — every sequence followed by one 0 delimiter */
    uint64_t n = seq_off[nseq] - seq_off[0] + nseq;
    if (n == 0 || sampling_rate == 0) return NULL;
    uint8_t* text = malloc(n);
    uint64_t* tseq = malloc(n * 8); uint64_t* tpos = malloc(n * 8);
    uint64_t w = 0;
    for (uint64_t s = 0; s < nseq; ++s) {
        uint64_t len = seq_off[s + 1] - seq_off[s];
        for (uint64_t j = 0; j <= len; ++j, ++w) {
            text[w] = j < len ? seqs[seq_off[s] + j] : 0;
            tseq[w] = s; tpos[w] = j;                         /* FMIndex.h:79-101: (refId, pos), delimiter included */
        }
    }
    uint64_t* sa = malloc(n * 8);
    uint8_t* bwt = malloc(n); uint8_t* bwt_rev = NULL;
    if (ora_suffix_array(text, n, sa) != 0) { free(text); free(tseq); free(tpos); free(sa); free(bwt); return NULL; }
    ora_bwt_from_sa(text, n, sa, bwt);
    uint8_t* has = malloc(n); uint64_t* vs = malloc(n * 8); uint64_t* vp = malloc(n * 8);
    for (uint64_t i = 0; i < n; ++i) {                        /* utils.h:236-240 */
        uint64_t p = sa[i];
        has[i] = tpos[p] % sampling_rate == 0;
        vs[i] = tseq[p]; vp[i] = tpos[p];
    }
    if (bidirectional) {                                      /* BiFMIndex.h:78-92: reverse whole text, second SA */
        uint8_t* rev = malloc(n);
        for (uint64_t i = 0; i < n; ++i) rev[i] = text[n - 1 - i];
        uint64_t* sar = malloc(n * 8);
        ora_suffix_array(rev, n, sar);
        bwt_rev = malloc(n);
        ora_bwt_from_sa(rev, n, sar, bwt_rev);
        free(rev); free(sar);
    }
    ora_index* x = ora_index_from_bwt(layout, sigma, bwt, bwt_rev, n, has, vs, vp);
    free(text); free(tseq); free(tpos); free(sa); free(bwt); free(bwt_rev); free(has); free(vs); free(vp);
    return x;
}

void ora_index_free(ora_index* x) {
    if (!x) return;
    ora_string_free(x->bwt); ora_string_free(x->bwt_rev); ora_sparse_free(x->sa); free(x);
}

ora_cursor ora_cursor_init(const ora_index* x) { ora_cursor c = {0, 0, x->n}; return c; }

ora_cursor ora_extend_left(const ora_index* x, ora_cursor c, uint64_t symb) {   /* BiFMIndexCursor.h:113-120, FMIndexCursor.h:33-37 */
    ora_cursor r;
    uint64_t a = ora_rank(x->bwt, c.lb, symb), b = ora_rank(x->bwt, c.lb + c.len, symb);
    r.lb = a + x->C[symb]; r.len = b - a;
    r.lb_rev = x->bidirectional ? c.lb_rev + ora_prefix_rank(x->bwt, c.lb + c.len, symb) - ora_prefix_rank(x->bwt, c.lb, symb) : 0;
    return r;
}
ora_cursor ora_extend_right(const ora_index* x, ora_cursor c, uint64_t symb) {  /* BiFMIndexCursor.h:121-128 */
    ora_cursor r;
    uint64_t a = ora_rank(x->bwt_rev, c.lb_rev, symb), b = ora_rank(x->bwt_rev, c.lb_rev + c.len, symb);
    r.lb_rev = a + x->C[symb]; r.len = b - a;
    r.lb = c.lb + ora_prefix_rank(x->bwt_rev, c.lb_rev + c.len, symb) - ora_prefix_rank(x->bwt_rev, c.lb_rev, symb);
    return r;
}
void ora_extend_left_all(const ora_index* x, ora_cursor c, ora_cursor* out) {   /* BiFMIndexCursor.h:58-69, FMIndexCursor.h:38-53 */
    uint64_t rs1[256], prs1[256], rs2[256], prs2[256];
    ora_all_ranks_and_prefix_ranks(x->bwt, c.lb, rs1, prs1);
    ora_all_ranks_and_prefix_ranks(x->bwt, c.lb + c.len, rs2, prs2);
    for (int i = 0; i < x->sigma; ++i) {
        out[i].lb = rs1[i] + x->C[i]; out[i].len = rs2[i] - rs1[i];
        out[i].lb_rev = x->bidirectional ? c.lb_rev + prs2[i] - prs1[i] : 0;
    }
}
void ora_extend_right_all(const ora_index* x, ora_cursor c, ora_cursor* out) {  /* BiFMIndexCursor.h:71-82 */
    uint64_t rs1[256], prs1[256], rs2[256], prs2[256];
    ora_all_ranks_and_prefix_ranks(x->bwt_rev, c.lb_rev, rs1, prs1);
    ora_all_ranks_and_prefix_ranks(x->bwt_rev, c.lb_rev + c.len, rs2, prs2);
    for (int i = 0; i < x->sigma; ++i) {
        out[i].lb_rev = rs1[i] + x->C[i]; out[i].len = rs2[i] - rs1[i];
        out[i].lb = c.lb + prs2[i] - prs1[i];
    }
}

void ora_locate(const ora_index* x, uint64_t row, uint64_t* seq, uint64_t* pos, uint64_t* steps) {   /* FMIndex.h:113-124 */
    uint64_t st = 0;
    while (!ora_sparse_value(x->sa, row, seq, pos)) {
        uint64_t symb = ora_symbol(x->bwt, row);
        row = ora_rank(x->bwt, row, symb) + x->C[symb];
        ++st;
    }
    *steps = st;
}

/* =====================================================================================
 * searches
 * ===================================================================================== */
void ora_search_exact(const ora_index* x, const uint8_t* qbuf, const uint64_t* qoff, uint64_t nq,
                      uint64_t* out_lb, uint64_t* out_len, uint64_t* out_steps, int nthreads) {
    /* search/SearchNoErrors.h:12-26 */
    (void)nthreads;
    #pragma omp parallel for schedule(dynamic, 1024) num_threads(nthreads > 0 ? nthreads : 1)
    for (int64_t q = 0; q < (int64_t)nq; ++q) {
        const uint8_t* query = qbuf + qoff[q];
        uint64_t m = qoff[q + 1] - qoff[q], steps = 0;
        uint64_t lb = 0, len = x->n;
        for (uint64_t i = 0; i < m; ++i) {
            uint64_t r = query[m - i - 1];
            uint64_t a = ora_rank(x->bwt, lb, r), b = ora_rank(x->bwt, lb + len, r);   /* FMIndexCursor.h:33-37 */
            lb = a + x->C[r]; len = b - a; ++steps;
            if (len == 0) break;
        }
        out_lb[q] = lb; out_len[q] = len;
        if (out_steps) out_steps[q] = steps;
    }
}

/* search/SearchNoErrors.h:28-86: up to `batch` cursors advanced round-robin, one symbol each per pass (the reference's way of keeping several
 * independent cache misses in flight); a query leaves the batch when its interval is empty or it is finished.  Same results as above. */
void ora_search_exact_batched(const ora_index* x, const uint8_t* qbuf, const uint64_t* qoff, uint64_t nq,
                              uint64_t* out_lb, uint64_t* out_len, int batch, int nthreads) {
    if (batch < 1) batch = 32;
    if (batch > 256) batch = 256;
    int T = nthreads > 0 ? nthreads : 1;
    const uint64_t chunk = 4096;
    #pragma omp parallel for schedule(dynamic, 1) num_threads(T)
    for (int64_t c0 = 0; c0 < (int64_t)((nq + chunk - 1) / chunk); ++c0) {
        const uint64_t first = (uint64_t)c0 * chunk, last = first + chunk < nq ? first + chunk : nq;
        uint64_t qi[256], lb[256], len[256], st[256];
        int live = 0;
        uint64_t next = first;
        for (;;) {
            while (live < batch && next < last) { qi[live] = next; lb[live] = 0; len[live] = x->n; st[live] = 0; ++live; ++next; }   /* fillBatch */
            if (live == 0) break;
            for (int k = 0; k < live; ++k) {                        /* doBatchJump: one extendLeft per cursor */
                const uint64_t q = qi[k], m = qoff[q + 1] - qoff[q];
                if (st[k] < m) {
                    const uint64_t r = qbuf[qoff[q] + m - st[k] - 1];
                    const uint64_t a = ora_rank(x->bwt, lb[k], r), b = ora_rank(x->bwt, lb[k] + len[k], r);
                    lb[k] = a + x->C[r]; len[k] = b - a; ++st[k];
                }
            }
            int w = 0;
            for (int k = 0; k < live; ++k) {                        /* remove_if: empty or finished */
                const uint64_t q = qi[k], m = qoff[q + 1] - qoff[q];
                if (len[k] == 0 || st[k] == m) { out_lb[q] = lb[k]; out_len[q] = len[k]; }
                else { qi[w] = qi[k]; lb[w] = lb[k]; len[w] = len[k]; st[w] = st[k]; ++w; }
            }
            live = w;
        }
    }
}

typedef struct emit_ctx { ora_hit* out; uint64_t cap, count, qidx, nodes; uint64_t quota; } emit_ctx;
static void emit(emit_ctx* e, ora_cursor c, uint64_t errors) {
    if (e->count < e->cap) {
        ora_hit* h = &e->out[e->count];
        h->qidx = e->qidx; h->lb = c.lb; h->lb_rev = c.lb_rev; h->len = c.len; h->errors = errors;
    }
    e->count++;
}

/* search/Backtracking.h:66-77 */
static void bt_no_errors(const ora_index* x, emit_ctx* e, const uint8_t* q, uint64_t m, ora_cursor cur, uint64_t i, uint64_t maxErrors) {
    if (cur.len == 0) return;
    for (; i < m; ++i) {
        cur = ora_extend_left(x, cur, q[m - i - 1]); e->nodes++;
        if (cur.len == 0) return;
    }
    emit(e, cur, maxErrors);
}
/* search/Backtracking.h:42-64 */
static void bt_with_errors(const ora_index* x, emit_ctx* ctx, const uint8_t* q, uint64_t m, uint64_t e, ora_cursor cur, uint64_t i, uint64_t maxErrors) {
    if (cur.len == 0) return;
    if (e == maxErrors) { bt_no_errors(x, ctx, q, m, cur, i, maxErrors); return; }
    ora_cursor kids[256];
    for (; i < m; ++i) {
        uint64_t r = q[m - i - 1];
        ora_extend_left_all(x, cur, kids); ctx->nodes++;
        for (uint64_t s = 1; s < (uint64_t)x->sigma; ++s) if (r != s) {
            ora_cursor child = kids[s];
            bt_with_errors(x, ctx, q, m, e + 1, child, i + 1, maxErrors);
        }
        cur = kids[r];
        if (cur.len == 0) return;
    }
    emit(ctx, cur, e);
}
uint64_t ora_search_backtracking(const ora_index* x, const uint8_t* qbuf, const uint64_t* qoff, uint64_t nq,
                                 uint64_t max_errors, ora_hit* out, uint64_t cap, uint64_t* out_nodes) {
    emit_ctx e = {out, cap, 0, 0, 0, UINT64_MAX};
    for (uint64_t q = 0; q < nq; ++q) {
        e.qidx = q;
        bt_with_errors(x, &e, qbuf + qoff[q], qoff[q + 1] - qoff[q], 0, ora_cursor_init(x), 0, max_errors);
    }
    if (out_nodes) *out_nodes = e.nodes;
    return e.count;
}

/* ---- search_ng26<Edit=false> (search/SearchNg26.h:18-366), SURVEY.md appendix A -------- */
typedef struct ng_search {
    const ora_index* x; const uint8_t* q; uint64_t m;
    int P; const uint64_t *pi, *l, *u, *part;
    emit_ctx* e;
} ng_search;
typedef struct ng_state { ora_cursor cur; uint64_t e, part, pev, qL, qR; int Right; } ng_state;

static int ng_next(const ng_search* s, ng_state st);
static int ng_dir(const ng_search* s, ng_state st);
static int ng_single(const ng_search* s, ng_state st);

static ora_cursor ng_extend(const ng_search* s, const ng_state* st, uint64_t c) {
    s->e->nodes++;
    return st->Right ? ora_extend_right(s->x, st->cur, c) : ora_extend_left(s->x, st->cur, c);
}
/* delegate with search_n clipping, SearchNg26.h:412-420 */
static int ng_report(const ng_search* s, ora_cursor cur, uint64_t e) {
    emit_ctx* c = s->e;
    if (cur.len > c->quota) cur.len = c->quota;
    c->quota -= cur.len;
    emit(c, cur, e);
    return c->quota == 0;
}
static int ng_next(const ng_search* s, ng_state st) {            /* search_next, :98-117 */
    if (st.cur.len == 0) return 0;
    if (st.part == (uint64_t)s->P) {
        if (s->l[s->P - 1] <= st.e && st.e <= s->u[s->P - 1]) return ng_report(s, st.cur, st.e);
        return 0;
    }
    st.Right = (st.part == 0) || (s->pi[st.part - 1] < s->pi[st.part]);
    return st.cur.len > 1 ? ng_dir(s, st) : ng_single(s, st);
}
static int ng_advance(const ng_search* s, ng_state st) {         /* search_next_pos with NextPos = true, :119-141 */
    if (st.cur.len == 0) return 0;
    if (st.Right) st.qR += 1; else st.qL -= 1;
    st.pev -= 1;
    if (st.pev == 0) {
        st.part += 1;
        if (st.part != (uint64_t)s->P) st.pev = s->part[s->pi[st.part]];
        return ng_next(s, st);
    }
    return st.cur.len > 1 ? ng_dir(s, st) : ng_single(s, st);
}
static int ng_exact_tail(const ng_search* s, ng_state st) {      /* search_next_dir_no_errors, :225-250 */
    uint64_t loops = st.pev;
    for (uint64_t i = 0; i < loops; ++i) {
        uint64_t c = s->q[st.Right ? st.qR + i : st.qL - i];
        st.cur = ng_extend(s, &st, c);
        if (st.cur.len == 0) return 0;
    }
    st.part += 1; st.pev = 0;
    if (st.part != (uint64_t)s->P) st.pev = s->part[s->pi[st.part]];
    if (st.Right) st.qR += loops; else st.qL -= loops;
    return ng_next(s, st);
}
static int ng_dir(const ng_search* s, ng_state st) {             /* search_next_dir, :143-224 */
    uint64_t c = s->q[st.Right ? st.qR : st.qL];
    int mOK = (st.pev > 1 || s->l[st.part] <= st.e) && st.e <= s->u[st.part];
    int sOK = (st.pev > 1 || s->l[st.part] <= st.e + 1) && st.e + 1 <= s->u[st.part];
    int xOK = st.e + 1 <= s->u[st.part];
    if (xOK) {
        ora_cursor kids[256];
        s->e->nodes++;
        if (st.Right) ora_extend_right_all(s->x, st.cur, kids); else ora_extend_left_all(s->x, st.cur, kids);
        if (mOK) { ng_state n = st; n.cur = kids[c]; if (ng_advance(s, n)) return 1; }
        for (uint64_t i = 1 /* FirstSymb, BiFMIndex.h:26 */; i < (uint64_t)s->x->sigma; ++i) {
            if (!sOK) continue;
            if (i == c) continue;
            ng_state n = st; n.e = st.e + 1; n.cur = kids[i];
            if (ng_advance(s, n)) return 1;
        }
    } else if (mOK) {
        if (ng_exact_tail(s, st)) return 1;
    }
    return 0;
}
static int ng_single(const ng_search* s, ng_state st) {          /* search_next_dir_single, :251-365 */
    uint64_t b = st.Right ? ora_symbol(s->x->bwt_rev, st.cur.lb_rev) : ora_symbol(s->x->bwt, st.cur.lb);   /* BiFMIndexCursor.h:180-190 */
    ora_cursor nx = ng_extend(s, &st, b);
    uint64_t c = s->q[st.Right ? st.qR : st.qL];
    int sOK = (st.pev > 1 || s->l[st.part] <= st.e + 1) && st.e + 1 <= s->u[st.part];
    int xOK = st.e + 1 <= s->u[st.part];
    if (b < 1 /* FirstSymb */) return 0;
    int mOK = (st.pev > 1 || s->l[st.part] <= st.e) && st.e <= s->u[st.part];
    if (b == c) {
        if (mOK) {
            if (!xOK) return ng_exact_tail(s, st);
            ng_state n = st; n.cur = nx;
            if (ng_advance(s, n)) return 1;
        }
    } else if (xOK) {
        if (sOK) { ng_state n = st; n.e = st.e + 1; n.cur = nx; if (ng_advance(s, n)) return 1; }
    }
    return 0;
}
static int ng_run(const ng_search* s) {                          /* run, :62-79 */
    ng_state st; memset(&st, 0, sizeof st);
    for (uint64_t i = 0; i < s->pi[0]; ++i) { st.qL += s->part[i]; st.qR += s->part[i]; }
    st.qL -= 1;
    st.pev = s->part[s->pi[0]];
    st.cur = ora_cursor_init(s->x);
    return ng_next(s, st);
}

uint64_t ora_search_ng26_hamming(const ora_index* x, const uint8_t* qbuf, const uint64_t* qoff, uint64_t nq,
                                 int nsearch, int nparts, const uint64_t* pi, const uint64_t* l, const uint64_t* u,
                                 const uint64_t* partition, uint64_t max_hits_per_query,
                                 ora_hit* out, uint64_t cap, uint64_t* out_qcount, uint64_t* out_nodes, int nthreads) {
    if (!x->bidirectional || nq == 0 || max_hits_per_query == 0) { if (out_nodes) *out_nodes = 0; return 0; }   /* SearchNg26.h:408-409 */
    uint64_t psum = 0;
    if (partition) for (int p = 0; p < nparts; ++p) psum += partition[p];
    int T = nthreads > 0 ? nthreads : 1;
    uint64_t total = 0, total_nodes = 0;
    if (T == 1) {
        emit_ctx e = {out, cap, 0, 0, 0, 0};
        for (uint64_t q = 0; q < nq; ++q) {                      /* search_n_impl, :407-423 */
            uint64_t before = e.count, part_buf[64];
            uint64_t m = qoff[q + 1] - qoff[q];
            const uint64_t* part = partition;
            if (m < (uint64_t)nparts || (partition && m != psum)) { if (out_qcount) out_qcount[q] = 0; continue; }   /* expand.h:325-327 asserts parts <= length, and an explicit partition must add up to the query: such a query is skipped */
            if (!part) { ora_uniform_partition((uint64_t)nparts, m, part_buf); part = part_buf; }
            e.qidx = q; e.quota = max_hits_per_query;
            for (int si = 0; si < nsearch; ++si) {               /* search_impl, :369-391 */
                ng_search s = {x, qbuf + qoff[q], m, nparts, pi + si * nparts, l + si * nparts, u + si * nparts, part, &e};
                if (ng_run(&s)) break;
            }
            if (out_qcount) out_qcount[q] = e.count - before;
        }
        total = e.count; total_nodes = e.nodes;
    } else {
        /* threaded variant for the CPU baseline: counts only (no hit records), same traversal */
        #pragma omp parallel for schedule(dynamic, 256) num_threads(T) reduction(+:total, total_nodes)
        for (int64_t q = 0; q < (int64_t)nq; ++q) {
            emit_ctx e = {NULL, 0, 0, (uint64_t)q, 0, max_hits_per_query};
            uint64_t part_buf[64];
            uint64_t m = qoff[q + 1] - qoff[q];
            const uint64_t* part = partition;
            if (m < (uint64_t)nparts || (partition && m != psum)) { if (out_qcount) out_qcount[q] = 0; continue; }   /* expand.h:325-327 asserts parts <= length, and an explicit partition must add up to the query: such a query is skipped */
            if (!part) { ora_uniform_partition((uint64_t)nparts, m, part_buf); part = part_buf; }
            for (int si = 0; si < nsearch; ++si) {
                ng_search s = {x, qbuf + qoff[q], m, nparts, pi + si * nparts, l + si * nparts, u + si * nparts, part, &e};
                if (ng_run(&s)) break;
            }
            if (out_qcount) out_qcount[q] = e.count;
            total += e.count; total_nodes += e.nodes;
        }
        /* hit records from the threaded variant (bench.py's parity check at full index size): the per-query counts give every query its slice
         * of `out`, a second pass over the queries fills the slices — same order as the single-threaded walk */
        if (out && out_qcount && total <= cap) {
            uint64_t* first = malloc((nq + 1) * 8);
            first[0] = 0;
            for (uint64_t q = 0; q < nq; ++q) first[q + 1] = first[q] + out_qcount[q];
            #pragma omp parallel for schedule(dynamic, 256) num_threads(T)
            for (int64_t q = 0; q < (int64_t)nq; ++q) {
                if (out_qcount[q] == 0) continue;
                emit_ctx e = {out + first[q], out_qcount[q], 0, (uint64_t)q, 0, max_hits_per_query};
                uint64_t part_buf[64];
                uint64_t m = qoff[q + 1] - qoff[q];
                const uint64_t* part = partition;
                if (!part) { ora_uniform_partition((uint64_t)nparts, m, part_buf); part = part_buf; }
                for (int si = 0; si < nsearch; ++si) {
                    ng_search s = {x, qbuf + qoff[q], m, nparts, pi + si * nparts, l + si * nparts, u + si * nparts, part, &e};
                    if (ng_run(&s)) break;
                }
            }
            free(first);
        }
    }
    if (out_nodes) *out_nodes = total_nodes;
    return total;
}

/* ---- search_ng26<Edit> (search/SearchNg26.h:18-366), the whole state machine incl. the Edit = true branches ----------
 * Separate from the Hamming reduction above on purpose: ng_* stays the pinned restatement of SURVEY appendix A, nge_* follows
 * the header line by line (State :41-53) and must agree with it for edit == 0 (tests/test_oracle_golden.py). */
typedef struct nge_state {
    ora_cursor cur;
    uint8_t lastRank[2], lastQRank[2];      /* side[0] = left, side[1] = right (:36-39, indexed by state.Right) */
    uint64_t e, part, pev, qL, qR;
    char LInfo, RInfo;
    int Right, NextPos;
} nge_state;
typedef struct nge_search { ng_search b; int edit; } nge_search;

static int nge_next(const nge_search* s, nge_state st);
static int nge_dir(const nge_search* s, nge_state st);
static int nge_single(const nge_search* s, nge_state st);

static ora_cursor nge_extend(const nge_search* s, const nge_state* st, uint64_t c) {
    s->b.e->nodes++;
    return st->Right ? ora_extend_right(s->b.x, st->cur, c) : ora_extend_left(s->b.x, st->cur, c);
}
static int nge_next(const nge_search* s, nge_state st) {          /* search_next, :98-117 */
    const ng_search* b = &s->b;
    if (st.cur.len == 0) return 0;
    if (st.part == (uint64_t)b->P) {
        if (!s->edit || ((st.LInfo == 'M' || st.LInfo == 'I') && (st.RInfo == 'M' || st.RInfo == 'I')))
            if (b->l[b->P - 1] <= st.e && st.e <= b->u[b->P - 1]) return ng_report(b, st.cur, st.e);
        return 0;
    }
    st.Right = (st.part == 0) || (b->pi[st.part - 1] < b->pi[st.part]);
    return st.cur.len > 1 ? nge_dir(s, st) : nge_single(s, st);
}
static int nge_pos(const nge_search* s, nge_state st) {           /* search_next_pos, :119-141 */
    const ng_search* b = &s->b;
    if (st.cur.len == 0) return 0;
    if (st.NextPos) {
        if (st.Right) st.qR += 1; else st.qL -= 1;
        st.pev -= 1;
        if (st.pev == 0) {
            st.part += 1;
            if (st.part != (uint64_t)b->P) st.pev = b->part[b->pi[st.part]];
            return nge_next(s, st);
        }
    }
    return st.cur.len > 1 ? nge_dir(s, st) : nge_single(s, st);
}
static int nge_no_errors(const nge_search* s, nge_state st) {     /* search_next_dir_no_errors, :225-250 */
    const ng_search* b = &s->b;
    uint64_t loops = st.pev, c = 0;
    for (uint64_t i = 0; i < loops; ++i) {
        c = b->q[st.Right ? st.qR + i : st.qL - i];
        st.cur = nge_extend(s, &st, c);
        if (st.cur.len == 0) return 0;
    }
    st.lastRank[st.Right] = (uint8_t)c; st.lastQRank[st.Right] = (uint8_t)c;
    st.part += 1; st.pev = 0;
    if (st.part != (uint64_t)b->P) st.pev = b->part[b->pi[st.part]];
    if (st.Right) { st.qR += loops; st.RInfo = 'M'; } else { st.qL -= loops; st.LInfo = 'M'; }
    return nge_next(s, st);
}
#define NGE_INFO(n, kind) do { if (st.Right) (n).RInfo = (kind); else (n).LInfo = (kind); } while (0)   /* OnMatchL/R ... :149-156 */
static int nge_dir(const nge_search* s, nge_state st) {           /* search_next_dir, :143-224 */
    const ng_search* b = &s->b;
    const char T = st.Right ? st.RInfo : st.LInfo;
    const int Deletion = (T != 'S' && T != 'I') && s->edit, Insertion = (T != 'S' && T != 'D') && s->edit;
    const uint64_t c = b->q[st.Right ? st.qR : st.qL];
    const int mOK = (st.pev > 1 || b->l[st.part] <= st.e) && st.e <= b->u[st.part]
                    && (T != 'I' || c != st.lastQRank[st.Right]) && (T != 'D' || c != st.lastRank[st.Right]);
    const int iOK = (st.pev > 1 || b->l[st.part] <= st.e + 1) && st.e + 1 <= b->u[st.part];
    const int sOK = iOK, xOK = st.e + 1 <= b->u[st.part];
    if (xOK) {
        ora_cursor kids[256];
        b->e->nodes++;
        if (st.Right) ora_extend_right_all(b->x, st.cur, kids); else ora_extend_left_all(b->x, st.cur, kids);
        if (mOK) {
            nge_state n = st; n.cur = kids[c]; n.lastRank[st.Right] = (uint8_t)c; n.lastQRank[st.Right] = (uint8_t)c;
            NGE_INFO(n, 'M'); n.NextPos = 1;
            if (nge_pos(s, n)) return 1;
        }
        for (uint64_t i = 1 /* FirstSymb */; i < (uint64_t)b->x->sigma; ++i) {
            nge_state n = st; n.e = st.e + 1; n.cur = kids[i]; n.lastRank[st.Right] = (uint8_t)i;
            if (Deletion) { NGE_INFO(n, 'D'); n.NextPos = 0; if (nge_pos(s, n)) return 1; }
            if (!sOK) continue;
            if (i == c) continue;
            n.lastQRank[st.Right] = (uint8_t)c; NGE_INFO(n, 'S'); n.NextPos = 1;
            if (nge_pos(s, n)) return 1;
        }
        if (Insertion && iOK) {
            nge_state n = st; n.e = st.e + 1; n.lastQRank[st.Right] = (uint8_t)c; NGE_INFO(n, 'I'); n.NextPos = 1;
            if (nge_pos(s, n)) return 1;
        }
    } else if (mOK) {
        if (nge_no_errors(s, st)) return 1;
    }
    return 0;
}
static int nge_single(const nge_search* s, nge_state st) {        /* search_next_dir_single, :251-365 */
    const ng_search* b = &s->b;
    const char T = st.Right ? st.RInfo : st.LInfo;
    const int Deletion = (T != 'S' && T != 'I') && s->edit, Insertion = (T != 'S' && T != 'D') && s->edit;
    const uint64_t bs = st.Right ? ora_symbol(b->x->bwt_rev, st.cur.lb_rev) : ora_symbol(b->x->bwt, st.cur.lb);
    const ora_cursor nx = nge_extend(s, &st, bs);
    const uint64_t c = b->q[st.Right ? st.qR : st.qL];
    const int iOK = (st.pev > 1 || b->l[st.part] <= st.e + 1) && st.e + 1 <= b->u[st.part];
    const int sOK = iOK, xOK = st.e + 1 <= b->u[st.part];
    if (Insertion && iOK) {
        nge_state n = st; n.e = st.e + 1; n.lastQRank[st.Right] = (uint8_t)c; NGE_INFO(n, 'I'); n.NextPos = 1;
        if (nge_pos(s, n)) return 1;
    }
    if (bs < 1 /* FirstSymb */) return 0;
    const int mOK = (st.pev > 1 || b->l[st.part] <= st.e) && st.e <= b->u[st.part]
                    && (T != 'I' || c != st.lastQRank[st.Right]) && (T != 'D' || c != st.lastRank[st.Right]);
    if (bs == c) {
        if (mOK) {
            if (!xOK) return nge_no_errors(s, st);
            nge_state n = st; n.lastRank[st.Right] = (uint8_t)c; n.lastQRank[st.Right] = (uint8_t)c; n.cur = nx;
            NGE_INFO(n, 'M'); n.NextPos = 1;
            if (nge_pos(s, n)) return 1;
        }
        if (Deletion && xOK) {
            nge_state n = st; n.e = st.e + 1; n.lastRank[st.Right] = (uint8_t)bs; n.cur = nx; NGE_INFO(n, 'D'); n.NextPos = 0;
            if (nge_pos(s, n)) return 1;
        }
    } else if (xOK) {
        nge_state n = st; n.e = st.e + 1; n.lastRank[st.Right] = (uint8_t)bs; n.cur = nx;
        if (sOK) {
            nge_state m = n; m.lastQRank[st.Right] = (uint8_t)c; NGE_INFO(m, 'S'); m.NextPos = 1;       /* Restore{} :341 */
            if (nge_pos(s, m)) return 1;
        }
        if (Deletion) { NGE_INFO(n, 'D'); n.NextPos = 0; if (nge_pos(s, n)) return 1; }
    }
    return 0;
}
static int nge_run(const nge_search* s) {                          /* run, :62-79 */
    const ng_search* b = &s->b;
    nge_state st; memset(&st, 0, sizeof st);
    for (uint64_t i = 0; i < b->pi[0]; ++i) { st.qL += b->part[i]; st.qR += b->part[i]; }
    st.qL -= 1;
    st.pev = b->part[b->pi[0]];
    st.cur = ora_cursor_init(b->x);
    st.LInfo = 'M'; st.RInfo = 'M';
    return nge_next(s, st);
}

uint64_t ora_search_ng26(const ora_index* x, int edit, const uint8_t* qbuf, const uint64_t* qoff, uint64_t nq,
                         int nsearch, int nparts, const uint64_t* pi, const uint64_t* l, const uint64_t* u,
                         const uint64_t* partition, uint64_t max_hits_per_query,
                         ora_hit* out, uint64_t cap, uint64_t* out_qcount, uint64_t* out_nodes, int nthreads) {
    if (!x->bidirectional || nq == 0 || max_hits_per_query == 0) { if (out_nodes) *out_nodes = 0; return 0; }
    uint64_t psum = 0;
    if (partition) for (int p = 0; p < nparts; ++p) psum += partition[p];
    int T = nthreads > 0 ? nthreads : 1;
    uint64_t total = 0, total_nodes = 0;
    if (T == 1) {
        emit_ctx e = {out, cap, 0, 0, 0, 0};
        for (uint64_t q = 0; q < nq; ++q) {
            uint64_t before = e.count, part_buf[64];
            uint64_t m = qoff[q + 1] - qoff[q];
            const uint64_t* part = partition;
            if (m < (uint64_t)nparts || (partition && m != psum)) { if (out_qcount) out_qcount[q] = 0; continue; }   /* expand.h:325-327 asserts parts <= length, and an explicit partition must add up to the query: such a query is skipped */
            if (!part) { ora_uniform_partition((uint64_t)nparts, m, part_buf); part = part_buf; }
            e.qidx = q; e.quota = max_hits_per_query;
            for (int si = 0; si < nsearch; ++si) {
                nge_search s = {{x, qbuf + qoff[q], m, nparts, pi + si * nparts, l + si * nparts, u + si * nparts, part, &e}, edit};
                if (nge_run(&s)) break;
            }
            if (out_qcount) out_qcount[q] = e.count - before;
        }
        total = e.count; total_nodes = e.nodes;
    } else {
        #pragma omp parallel for schedule(dynamic, 64) num_threads(T) reduction(+:total, total_nodes)
        for (int64_t q = 0; q < (int64_t)nq; ++q) {
            emit_ctx e = {NULL, 0, 0, (uint64_t)q, 0, max_hits_per_query};
            uint64_t part_buf[64];
            uint64_t m = qoff[q + 1] - qoff[q];
            const uint64_t* part = partition;
            if (m < (uint64_t)nparts || (partition && m != psum)) { if (out_qcount) out_qcount[q] = 0; continue; }   /* expand.h:325-327 asserts parts <= length, and an explicit partition must add up to the query: such a query is skipped */
            if (!part) { ora_uniform_partition((uint64_t)nparts, m, part_buf); part = part_buf; }
            for (int si = 0; si < nsearch; ++si) {
                nge_search s = {{x, qbuf + qoff[q], m, nparts, pi + si * nparts, l + si * nparts, u + si * nparts, part, &e}, edit};
                if (nge_run(&s)) break;
            }
            if (out_qcount) out_qcount[q] = e.count;
            total += e.count; total_nodes += e.nodes;
        }
    }
    if (out_nodes) *out_nodes = total_nodes;
    return total;
}

/* ---- search_ng21 (search/SearchNg21.h:26-156): edit-distance search over an EXPANDED scheme (one {pi, l, u} entry per query symbol,
 * search_scheme/expand.h:146-165).  No count()==1 path, one lastRank for both sides, the previous block's symbol instead of lastQRank,
 * the lower bound applied at every symbol, substitution / deletion children only for symbols 1..Sigma-1 other than the query's. ------- */
typedef struct n21_search {
    const ora_index* x; const uint8_t* q; uint64_t len;
    const uint64_t *pi, *l, *u;
    emit_ctx* e;
} n21_search;
static int n21_right(const n21_search* s, uint64_t k) {           /* prepare_reorder, :184-200 (a one-symbol search reads pi[1]: taken as Right here) */
    if (k == 0) return s->len < 2 || s->pi[0] < s->pi[1];
    return s->pi[k - 1] < s->pi[k];
}
static int n21_report(const n21_search* s, ora_cursor cur, uint64_t e) {   /* search_n, :223-240 (search: quota = UINT64_MAX, never full) */
    emit_ctx* c = s->e;
    if (cur.len > c->quota) cur.len = c->quota;
    c->quota -= cur.len;
    emit(c, cur, e);
    return c->quota == 0;
}
static int n21_next(const n21_search* s, ora_cursor cur, uint64_t e, uint64_t k, uint64_t lastRank, char LInfo, char RInfo) {   /* search_next + search_next_dir, :68-153 */
    if (cur.len == 0) return 0;
    if (k == s->len) {
        if ((LInfo == 'M' || LInfo == 'I') && (RInfo == 'M' || RInfo == 'I')) return n21_report(s, cur, e);
        return 0;
    }
    const int Right = n21_right(s, k);
    const char TInfo = Right ? RInfo : LInfo;
    const int Deletion = TInfo == 'M' || TInfo == 'D', Insertion = TInfo == 'M' || TInfo == 'I';
    const uint64_t symb = s->q[s->pi[k]];
    const int matchAllowed = s->l[k] <= e && e <= s->u[k]
                             && (TInfo != 'I' || symb != s->q[s->pi[k - 1]])
                             && (TInfo != 'D' || symb != lastRank);
    const int mismatchAllowed = s->l[k] <= e + 1 && e + 1 <= s->u[k];
    #define N21_L(op) (Right ? LInfo : (op))
    #define N21_R(op) (Right ? (op) : RInfo)
    if (mismatchAllowed) {
        ora_cursor kids[256];
        if (Right) ora_extend_right_all(s->x, cur, kids); else ora_extend_left_all(s->x, cur, kids);
        s->e->nodes++;
        if (matchAllowed && n21_next(s, kids[symb], e, k + 1, symb, N21_L('M'), N21_R('M'))) return 1;
        for (uint64_t i = 1; i < (uint64_t)s->x->sigma; ++i) {   /* :122-143, the two loops around symb */
            if (i == symb) continue;
            if (Deletion && n21_next(s, kids[i], e + 1, k, i, N21_L('D'), N21_R('D'))) return 1;
            if (n21_next(s, kids[i], e + 1, k + 1, i, N21_L('S'), N21_R('S'))) return 1;
        }
        if (Insertion && n21_next(s, cur, e + 1, k + 1, lastRank, N21_L('I'), N21_R('I'))) return 1;
    } else if (matchAllowed) {
        ora_cursor c = Right ? ora_extend_right(s->x, cur, symb) : ora_extend_left(s->x, cur, symb);
        s->e->nodes++;
        if (n21_next(s, c, e, k + 1, symb, N21_L('M'), N21_R('M'))) return 1;
    }
    #undef N21_L
    #undef N21_R
    return 0;
}
/* search / search_n, :205-240: pi, l, u hold nsearch rows of `len` entries (the expanded scheme); a query shorter than `len` would be read
 * out of bounds by the reference and is skipped; hits carry the query index, in callback order */
uint64_t ora_search_ng21(const ora_index* x, const uint8_t* qbuf, const uint64_t* qoff, uint64_t nq,
                         int nsearch, uint64_t len, const uint64_t* pi, const uint64_t* l, const uint64_t* u,
                         uint64_t max_hits_per_query, ora_hit* out, uint64_t cap, uint64_t* out_qcount, uint64_t* out_nodes) {
    if (out_nodes) *out_nodes = 0;
    if (!x->bidirectional || nq == 0 || nsearch <= 0) return 0;
    emit_ctx e = {out, cap, 0, 0, 0, 0};
    for (uint64_t q = 0; q < nq; ++q) {
        const uint64_t before = e.count, m = qoff[q + 1] - qoff[q];
        if (out_qcount) out_qcount[q] = 0;
        if (m < len) continue;
        e.qidx = q; e.quota = max_hits_per_query;
        for (int si = 0; si < nsearch; ++si) {
            n21_search s = {x, qbuf + qoff[q], len, pi + (uint64_t)si * len, l + (uint64_t)si * len, u + (uint64_t)si * len, &e};
            if (n21_next(&s, ora_cursor_init(x), 0, 0, 0, 'M', 'M')) break;
        }
        if (out_qcount) out_qcount[q] = e.count - before;
    }
    if (out_nodes) *out_nodes = e.nodes;
    return e.count;
}

/* =====================================================================================
 * search schemes
 * ===================================================================================== */
void ora_uniform_partition(uint64_t parts, uint64_t total, uint64_t* out) {   /* expand.h:324-335 */
    for (uint64_t i = 0; i < parts; ++i) out[i] = total / parts + (i < total % parts ? 1 : 0);
}

int ora_scheme_backtracking(uint64_t N, uint64_t minK, uint64_t K, uint64_t* pi, uint64_t* l, uint64_t* u) {   /* generator/backtracking.h:14-21 */
    for (uint64_t i = 0; i < N; ++i) { pi[i] = i; l[i] = 0; u[i] = K; }
    l[N - 1] = minK;
    return 1;
}

static int pigeon(uint64_t minK, uint64_t K, uint64_t* pi, uint64_t* l, uint64_t* u, int opt) {   /* generator/pigeon.h:14-102 */
    uint64_t N = K + 1;
    for (uint64_t i = 0; i < N; ++i) {
        uint64_t* P = pi + i * N; uint64_t* L = l + i * N; uint64_t* U = u + i * N;
        uint64_t k = 0;
        P[k] = i; L[k] = 0; U[k] = 0; ++k;
        for (uint64_t j = i; j > 0; --j, ++k) { P[k] = j - 1; L[k] = opt ? i - j + 1 : 0; U[k] = opt ? K - j + 1 : K; }
        for (uint64_t j = i + 1; j < N; ++j, ++k) { P[k] = j; L[k] = opt ? i : 0; U[k] = K; }
        if (L[N - 1] < minK) L[N - 1] = minK;
    }
    return (int)N;
}
int ora_scheme_pigeon_opt(uint64_t minK, uint64_t K, uint64_t* pi, uint64_t* l, uint64_t* u) { return pigeon(minK, K, pi, l, u, 1); }
int ora_scheme_pigeon_trivial(uint64_t minK, uint64_t K, uint64_t* pi, uint64_t* l, uint64_t* u) { return pigeon(minK, K, pi, l, u, 0); }

/* generator/h2.h:15-153 */
static uint64_t h2_pi(uint64_t row, uint64_t n, uint64_t N, uint64_t K) {    /* h2.h:17-27 with Mod = 0 */
    row = K - row;
    if (n < N - row) return n + row;
    return N - n - 1;
}
int ora_scheme_h2(uint64_t N, uint64_t minK, uint64_t K, uint64_t* pi, uint64_t* l, uint64_t* u) {
    uint64_t R = K + 1;
    uint64_t* diffs = calloc(R * N, 8);
    /* generateDiffMatrix, h2.h:39-54 */
    for (uint64_t i = K; i < N; ++i) for (uint64_t row = 0; row < R; ++row) diffs[row * N + i] = K - row;
    for (uint64_t i = 0; i < K; ++i) {
        for (uint64_t row = 0; row < K; ++row) diffs[row * N + i] = (row - i + K) % K;
        diffs[K * N + i] = K;
    }
    /* generateOptimizedDiffMatrix, h2.h:56-99 */
    #define MAT(r, c) diffs[(r) * N + (c)]
    for (uint64_t i = 0; i < N; ++i) {
        for (uint64_t j = 0; j < R; ++j) {
            if (i == j || MAT(j, i) == 0) continue;
            /* isValid(row=j, n=i, v) */
            #define H2_VALID(row, nn, v, res) do { res = 1; \
                if ((row) == (nn)) res = 0; \
                else if ((row) > (nn)) { for (uint64_t q_ = 0; q_ < (nn); ++q_) if (MAT(row, q_) < (v)) { res = 0; break; } } \
                else { for (uint64_t q_ = (row) + 1; q_ < (nn); ++q_) if (MAT(row, q_) > (v)) { res = 0; break; } } } while (0)
            int ok; H2_VALID(j, i, MAT(j, i), ok);
            if (!ok) {
                for (uint64_t k = j + 1; k < R; ++k) {
                    int a, b; H2_VALID(j, i, MAT(k, i), a); H2_VALID(k, i, MAT(j, i), b);
                    if (a && b) { uint64_t t = MAT(k, i); MAT(k, i) = MAT(j, i); MAT(j, i) = t; break; }
                }
            }
        }
    }
    /* pieces + lower bound, h2.h:29-37, :101-109 */
    memset(l, 0, R * N * 8); memset(u, 0, R * N * 8);
    for (uint64_t row = 0; row < R; ++row) for (uint64_t i = 0; i < N; ++i) pi[row * N + i] = h2_pi(row, i, N, K);
    for (uint64_t i = 0; i <= K; ++i) for (uint64_t j = 0; j < K - i + 1; ++j) l[i * N + (N - j - 1)] = i;
    /* upper bound, h2.h:111-126 */
    for (uint64_t i = 1; i < N; ++i)
        for (uint64_t row = R; row-- > 0;) {
            uint64_t j = pi[row * N + i];
            uint64_t a = u[row * N + i - 1], b = l[row * N + i - 1] + MAT(K - row, j);
            u[row * N + i] = a > b ? a : b;
        }
    #undef H2_VALID
    #undef MAT
    for (uint64_t row = 0; row < R; ++row) if (l[row * N + N - 1] < minK) l[row * N + N - 1] = minK;   /* h2.h:143-145 */
    free(diffs);
    return (int)R;
}

int ora_scheme_is_valid(int nsearch, uint64_t parts, const uint64_t* pi, const uint64_t* l, const uint64_t* u) {   /* isValid.h:13-93 */
    if (parts == 0) return nsearch == 0;
    for (int s = 0; s < nsearch; ++s) {
        const uint64_t *P = pi + s * parts, *L = l + s * parts, *U = u + s * parts;
        uint64_t lo = P[0], hi = P[0];
        for (uint64_t i = 1; i < parts; ++i) {
            if (P[i] == hi + 1) hi = P[i];
            else if (P[i] + 1 == lo) lo = P[i];
            else return 0;
        }
        if (lo != 0) return 0;
        for (uint64_t i = 1; i < parts; ++i) if (L[i - 1] > L[i] || U[i - 1] > U[i]) return 0;
        for (uint64_t i = 0; i < parts; ++i) if (L[i] > U[i]) return 0;
    }
    return 1;
}

static int covers(int nsearch, uint64_t parts, const uint64_t* pi, const uint64_t* l, const uint64_t* u, const uint8_t* cfg) {   /* isComplete.h:18-38 */
    for (int s = 0; s < nsearch; ++s) {
        uint64_t a = 0; int ok = 1;
        for (uint64_t i = 0; i < parts; ++i) {
            a += cfg[pi[s * parts + i]];
            if (!(l[s * parts + i] <= a && a <= u[s * parts + i])) { ok = 0; break; }
        }
        if (ok) return 1;
    }
    return 0;
}
static int complete_rec(int nsearch, uint64_t parts, const uint64_t* pi, const uint64_t* l, const uint64_t* u,
                        uint8_t* cfg, uint64_t k, uint64_t start, uint64_t minK, uint64_t maxK) {   /* isComplete.h:40-64 */
    if (k >= maxK) return 1;
    for (uint64_t i = start; i < parts; ++i) {
        cfg[i] += 1;
        if (k + 1 >= minK && !covers(nsearch, parts, pi, l, u, cfg)) { cfg[i] -= 1; return 0; }
        if (!complete_rec(nsearch, parts, pi, l, u, cfg, k + 1, i, minK, maxK)) { cfg[i] -= 1; return 0; }
        cfg[i] -= 1;
    }
    return 1;
}
int ora_scheme_is_complete(int nsearch, uint64_t parts, const uint64_t* pi, const uint64_t* l, const uint64_t* u, uint64_t minK, uint64_t maxK) {
    if (nsearch == 0) return 0;
    uint8_t* cfg = calloc(parts, 1);
    int ok = 1;
    if (minK == 0 && !covers(nsearch, parts, pi, l, u, cfg)) ok = 0;
    if (ok) ok = complete_rec(nsearch, parts, pi, l, u, cfg, 0, 0, minK, maxK);
    free(cfg);
    return ok;
}

/* expand.h:22-165 */
int ora_scheme_expand(int nsearch, uint64_t parts, const uint64_t* pi, const uint64_t* l, const uint64_t* u,
                      uint64_t newLen, uint64_t* opi, uint64_t* ol, uint64_t* ou) {
    uint64_t* counts = malloc(parts * 8); uint64_t* starts = malloc(parts * 8);
    ora_uniform_partition(parts, newLen, counts);                     /* expandCount, expand.h:37-49 */
    starts[0] = 0; for (uint64_t i = 1; i < parts; ++i) starts[i] = starts[i - 1] + counts[i - 1];
    int kept = 0;
    for (int s = 0; s < nsearch; ++s) {
        const uint64_t *P = pi + s * parts, *L = l + s * parts, *U = u + s * parts;
        uint64_t *OP = opi + kept * newLen, *OL = ol + kept * newLen, *OU = ou + kept * newLen;
        uint64_t kp = 0, kl = 0, ku = 0;
        for (uint64_t i = 0; i < parts; ++i) {
            /* forwards(), expand.h:22-29 */
            int fwd = i == 0 ? (parts == 1 || P[1] > P[0]) : (P[i] > P[i - 1]);
            uint64_t lo = starts[P[i]], cnt = counts[P[i]];
            if (fwd) for (uint64_t j = 0; j < cnt; ++j) OP[kp++] = lo + j;
            else     for (uint64_t j = cnt; j > 0; --j) OP[kp++] = lo + j - 1;
            /* expandLowerBound, expand.h:107-124 */
            uint64_t c = cnt;
            while (c > 1) { --c; OL[kl++] = i > 0 ? L[i - 1] : 0; }
            if (c > 0) OL[kl++] = L[i]; else if (kl > 0) OL[kl - 1] = L[i];
            /* expandUpperBound, expand.h:131-142 */
            for (uint64_t j = 0; j < cnt; ++j) OU[ku++] = U[i];
        }
        if (kp == newLen && ora_scheme_is_valid(1, newLen, OP, OL, OU)) ++kept;
    }
    free(counts); free(starts);
    return kept;
}

void ora_scheme_limit_to_hamming(int nsearch, uint64_t parts, uint64_t* l, uint64_t* u) {   /* expand.h:301-319 */
    for (int s = 0; s < nsearch; ++s) {
        uint64_t *L = l + s * parts, *U = u + s * parts;
        for (uint64_t i = parts - 1; i > 0; --i) {
            if (L[i] == 0) break;
            if (L[i - 1] < L[i] - 1) L[i - 1] = L[i] - 1;
        }
        for (uint64_t i = 1; i < parts; ++i) if (U[i] > U[i - 1] + 1) U[i] = U[i - 1] + 1;
    }
}

double ora_scheme_node_count_hamming(int nsearch, uint64_t parts, const uint64_t* l, const uint64_t* u, uint64_t sigma) {   /* nodeCount.h:19-57 */
    long double total = 0;
    for (int s = 0; s < nsearch; ++s) {
        const uint64_t *L = l + s * parts, *U = u + s * parts;
        uint64_t e = 0; for (uint64_t i = 0; i < parts; ++i) if (U[i] > e) e = U[i];
        long double* last = calloc(e + 1, sizeof(long double)); long double* cur = calloc(e + 1, sizeof(long double));
        last[0] = 1;
        long double acc = 0;
        for (uint64_t n = 1; n <= parts; ++n) {
            for (uint64_t i = 0; i <= e; ++i) {
                if (L[n - 1] <= i && i <= U[n - 1]) {
                    cur[i] = last[i];
                    if (i > 0) cur[i] += (long double)(sigma - 1) * last[i - 1];
                    acc += cur[i];
                } else cur[i] = 0;
            }
            long double* t = cur; cur = last; last = t;
        }
        total += acc;
        free(last); free(cur);
    }
    return (double)total;
}
