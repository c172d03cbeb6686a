/*
 * hd_cavlc.c — CAVLC residual block parsing (H.264 9.2: coeff_token, trailing ones, levels,
 * total_zeros, run_before).
 *
 * Tables are entered in the specification's own form — Table 9-5 / 9-7 / 9-8 / 9-9 / 9-10 as
 * (code length, code value) pairs — and compiled at start-up into two-level 8+8-bit lookup tables
 * by a generic prefix-code builder.  Output goes straight to its final place: coefficient k of the
 * zig-zag scan is written at its raster position, which is the order the frame job carries.
 *
 * Replaces (behaviourally) the reference's src/h264bsd_cavlc.c:749-916; the reference's
 * "level_prefix above 15 is an error" rule for baseline streams (:820-834) is kept.
 */
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include "hostdec.h"

/* ---- Table 9-5, indexed [table][4*total_coeff + trailing_ones]; table 0: 0<=nC<2, 1: 2<=nC<4,
 *      2: 4<=nC<8, 3: nC>=8 (6-bit fixed length) ---- */
static const uint8_t ct_len[4][68] = {
  { 1, 0, 0, 0,   6, 2, 0, 0,   8, 6, 3, 0,   9, 8, 7, 5,  10, 9, 8, 6,  11,10, 9, 7,  13,11,10, 8,
   13,13,11, 9,  13,13,13,10,  14,14,13,11,  14,14,14,13,  15,15,14,14,  15,15,15,14,  16,15,15,15,
   16,16,16,15,  16,16,16,16,  16,16,16,16 },
  { 2, 0, 0, 0,   6, 2, 0, 0,   6, 5, 3, 0,   7, 6, 6, 4,   8, 6, 6, 4,   8, 7, 7, 5,   9, 8, 8, 6,
   11, 9, 9, 6,  11,11,11, 7,  12,11,11, 9,  12,12,12,11,  12,12,12,11,  13,13,13,12,  13,13,13,13,
   13,14,13,13,  14,14,14,13,  14,14,14,14 },
  { 4, 0, 0, 0,   6, 4, 0, 0,   6, 5, 4, 0,   6, 5, 5, 4,   7, 5, 5, 4,   7, 5, 5, 4,   7, 6, 6, 4,
    7, 6, 6, 4,   8, 7, 7, 5,   8, 8, 7, 6,   9, 8, 8, 7,   9, 9, 8, 8,   9, 9, 9, 8,  10, 9, 9, 9,
   10,10,10,10,  10,10,10,10,  10,10,10,10 },
  { 6, 0, 0, 0,   6, 6, 0, 0,   6, 6, 6, 0,   6, 6, 6, 6,   6, 6, 6, 6,   6, 6, 6, 6,   6, 6, 6, 6,
    6, 6, 6, 6,   6, 6, 6, 6,   6, 6, 6, 6,   6, 6, 6, 6,   6, 6, 6, 6,   6, 6, 6, 6,   6, 6, 6, 6,
    6, 6, 6, 6,   6, 6, 6, 6,   6, 6, 6, 6 },
};
static const uint8_t ct_code[4][68] = {
  { 1, 0, 0, 0,   5, 1, 0, 0,   7, 4, 1, 0,   7, 6, 5, 3,   7, 6, 5, 3,   7, 6, 5, 4,  15, 6, 5, 4,
   11,14, 5, 4,   8,10,13, 4,  15,14, 9, 4,  11,10,13,12,  15,14, 9,12,  11,10,13, 8,  15, 1, 9,12,
   11,14,13, 8,   7,10, 9,12,   4, 6, 5, 8 },
  { 3, 0, 0, 0,  11, 2, 0, 0,   7, 7, 3, 0,   7,10, 9, 5,   7, 6, 5, 4,   4, 6, 5, 6,   7, 6, 5, 8,
   15, 6, 5, 4,  11,14,13, 4,  15,10, 9, 4,  11,14,13,12,   8,10, 9, 8,  15,14,13,12,  11,10, 9,12,
    7,11, 6, 8,   9, 8,10, 1,   7, 6, 5, 4 },
  {15, 0, 0, 0,  15,14, 0, 0,  11,15,13, 0,   8,12,14,12,  15,10,11,11,  11, 8, 9,10,   9,14,13, 9,
    8,10, 9, 8,  15,14,13,13,  11,14,10,12,  15,10,13,12,  11,14, 9,12,   8,10,13, 8,  13, 7, 9,12,
    9,12,11,10,   5, 8, 7, 6,   1, 4, 3, 2 },
  { 3, 0, 0, 0,   0, 1, 0, 0,   4, 5, 6, 0,   8, 9,10,11,  12,13,14,15,  16,17,18,19,  20,21,22,23,
   24,25,26,27,  28,29,30,31,  32,33,34,35,  36,37,38,39,  40,41,42,43,  44,45,46,47,  48,49,50,51,
   52,53,54,55,  56,57,58,59,  60,61,62,63 },
};
/* chroma DC (nC == -1), [4*total_coeff + trailing_ones], total_coeff 0..4 */
static const uint8_t cdc_len[20]  = { 2,0,0,0,  6,1,0,0,  6,6,3,0,  6,7,7,6,  6,8,8,7 };
static const uint8_t cdc_code[20] = { 1,0,0,0,  7,1,0,0,  4,6,1,0,  3,3,2,5,  2,3,2,0 };

/* ---- Tables 9-7 / 9-8: total_zeros for 4x4 blocks, [total_coeff-1][total_zeros] ---- */
static const uint8_t tz_len[15][16] = {
  {1,3,3,4,4,5,5,6,6,7,7,8,8,9,9,9}, {3,3,3,3,3,4,4,4,4,5,5,6,6,6,6,0}, {4,3,3,3,4,4,3,3,4,5,5,6,5,6,0,0},
  {5,3,4,4,3,3,3,4,3,4,5,5,5,0,0,0}, {4,4,4,3,3,3,3,3,4,5,4,5,0,0,0,0}, {6,5,3,3,3,3,3,3,4,3,6,0,0,0,0,0},
  {6,5,3,3,3,2,3,4,3,6,0,0,0,0,0,0}, {6,4,5,3,2,2,3,3,6,0,0,0,0,0,0,0}, {6,6,4,2,2,3,2,5,0,0,0,0,0,0,0,0},
  {5,5,3,2,2,2,4,0,0,0,0,0,0,0,0,0}, {4,4,3,3,1,3,0,0,0,0,0,0,0,0,0,0}, {4,4,2,1,3,0,0,0,0,0,0,0,0,0,0,0},
  {3,3,1,2,0,0,0,0,0,0,0,0,0,0,0,0}, {2,2,1,0,0,0,0,0,0,0,0,0,0,0,0,0}, {1,1,0,0,0,0,0,0,0,0,0,0,0,0,0,0},
};
static const uint8_t tz_code[15][16] = {
  {1,3,2,3,2,3,2,3,2,3,2,3,2,3,2,1}, {7,6,5,4,3,5,4,3,2,3,2,3,2,1,0,0}, {5,7,6,5,4,3,4,3,2,3,2,1,1,0,0,0},
  {3,7,5,4,6,5,4,3,3,2,2,1,0,0,0,0}, {5,4,3,7,6,5,4,3,2,1,1,0,0,0,0,0}, {1,1,7,6,5,4,3,2,1,1,0,0,0,0,0,0},
  {1,1,5,4,3,3,2,1,1,0,0,0,0,0,0,0}, {1,1,1,3,3,2,2,1,0,0,0,0,0,0,0,0}, {1,0,1,3,2,1,1,1,0,0,0,0,0,0,0,0},
  {1,0,1,3,2,1,1,0,0,0,0,0,0,0,0,0}, {0,1,1,2,1,3,0,0,0,0,0,0,0,0,0,0}, {0,1,1,1,1,0,0,0,0,0,0,0,0,0,0,0},
  {0,1,1,1,0,0,0,0,0,0,0,0,0,0,0,0}, {0,1,1,0,0,0,0,0,0,0,0,0,0,0,0,0}, {0,1,0,0,0,0,0,0,0,0,0,0,0,0,0,0},
};
/* Table 9-9(a): total_zeros for chroma DC 2x2, [total_coeff-1][total_zeros] */
static const uint8_t ctz_len[3][4]  = { {1,2,3,3}, {1,2,2,0}, {1,1,0,0} };
static const uint8_t ctz_code[3][4] = { {1,1,1,0}, {1,1,0,0}, {1,0,0,0} };
/* Table 9-10: run_before, [min(zerosLeft,7)-1][run_before] */
static const uint8_t rb_len[7][15] = {
  {1,1}, {1,2,2}, {2,2,2,2}, {2,2,2,3,3}, {2,2,3,3,3,3}, {2,3,3,3,3,3,3}, {3,3,3,3,3,3,3,4,5,6,7,8,9,10,11},
};
static const uint8_t rb_code[7][15] = {
  {1,0}, {1,1,0}, {3,2,1,0}, {3,2,1,1,0}, {3,2,3,2,1,0}, {3,0,1,3,2,5,4}, {7,6,5,4,3,2,1,1,1,1,1,1,1,1,1},
};

/* ---- generic two-level prefix-code lookup: 8 bits, then 8 more ---- */
typedef struct Vlc {
    int16_t  l1[256];        /* >=0: (len<<8)|sym ; <0: -(subtable+1) ; 0x7FFF: invalid */
    int16_t  l2[12][256];
    int      n_sub;
} Vlc;
#define VLC_BAD 0x7FFF

static void vlc_build(Vlc *v, const uint8_t *len, const uint8_t *code, int n)
{
    for (int i = 0; i < 256; i++) v->l1[i] = VLC_BAD;
    v->n_sub = 0;
    for (int s = 0; s < n; s++) {
        int L = len[s];
        if (!L) continue;
        uint32_t c = code[s];
        if (L <= 8) {
            uint32_t base = c << (8 - L);
            for (uint32_t k = 0; k < (1u << (8 - L)); k++) v->l1[base + k] = (int16_t)((L << 8) | s);
        } else {
            uint32_t hi = c >> (L - 8);      /* first 8 bits (code values fit 8 bits => mostly zeros) */
            int sub;
            if (v->l1[hi] == VLC_BAD) {
                sub = v->n_sub++;
                for (int i = 0; i < 256; i++) v->l2[sub][i] = VLC_BAD;
                v->l1[hi] = (int16_t)(-(sub + 1));
            } else sub = -v->l1[hi] - 1;
            uint32_t lo = (c & ((1u << (L - 8)) - 1)) << (16 - L);
            for (uint32_t k = 0; k < (1u << (16 - L)); k++) v->l2[sub][lo + k] = (int16_t)((L << 8) | s);
        }
    }
}
/* returns symbol index or -1; consumes bits */
static inline int vlc_get(const Vlc *v, BitReader *br)
{
    uint32_t w = br_peek32(br);
    int e = v->l1[w >> 24];
    if (e < 0) e = v->l2[-e - 1][(w >> 16) & 0xFF];
    if (e == VLC_BAD) return -1;
    br_skip(br, (uint32_t)e >> 8);
    return e & 0xFF;
}

static Vlc vlc_ct[3], vlc_cdc, vlc_tz[15], vlc_ctz[3], vlc_rb[7];
int hd_trace;
int hd_no_fast_skip;         /* HD_NO_FAST_SKIP in the environment (read with HD_TRACE): hd_mb.c takes the general path for everything */

/* built exactly once per process however many threads create decoder instances at the same time */
static void cavlc_build_tables(void)
{
    const char *t = getenv("HD_TRACE");
    hd_trace = t && *t && *t != '0';
    { const char *f = getenv("HD_NO_FAST_SKIP"); hd_no_fast_skip = f && *f && *f != '0'; }
    for (int t = 0; t < 3; t++) vlc_build(&vlc_ct[t], ct_len[t], ct_code[t], 68);
    vlc_build(&vlc_cdc, cdc_len, cdc_code, 20);
    for (int t = 0; t < 15; t++) vlc_build(&vlc_tz[t], tz_len[t], tz_code[t], 16);
    for (int t = 0; t < 3; t++) vlc_build(&vlc_ctz[t], ctz_len[t], ctz_code[t], 4);
    for (int t = 0; t < 7; t++) vlc_build(&vlc_rb[t], rb_len[t], rb_code[t], 15);
}

void hd_cavlc_init(void)
{
    static pthread_once_t once = PTHREAD_ONCE_INIT;
    pthread_once(&once, cavlc_build_tables);
}

static const uint8_t zigzag4x4[16] = { 0, 1, 4, 8, 5, 2, 3, 6, 9, 12, 13, 10, 7, 11, 14, 15 };

int hd_cavlc_block(BitReader *br, int nc, int max_coeff, int16_t *coef, int *spill)
{
    uint32_t unused = 0;
    return hd_cavlc_block_sum(br, nc, max_coeff, coef, spill, &unused);
}

/* ... and adds the magnitudes of the levels it stored in the block to *abs_sum (hd_mb.c: the residual-range bound of the
 * macroblock falls out of the parse instead of a second pass over the coefficients) */
/* A window of the bit stream kept in registers for the duration of one block: bits [pos, pos + avail) of the stream are the
 * top `avail` bits of win.  One 8-byte load per ~32 consumed bits instead of one per syntax element (BitReader's br_peek32
 * assembles its word from memory every time); the position goes back to the BitReader when the block is done.  Reading past
 * the end of the data yields zeros — the caller fails the block for the overrun in any case — and never touches memory behind
 * the 8 pad bytes. */
typedef struct { const uint8_t *buf; uint64_t win; uint32_t pos, size_bits; int avail; } BitWin;
static inline void bw_refill(BitWin *w)
{
    if (w->pos > w->size_bits) { w->win = 0; w->avail = 64; return; }
    const uint8_t *p = w->buf + (w->pos >> 3);
    const uint64_t v = ((uint64_t)p[0] << 56) | ((uint64_t)p[1] << 48) | ((uint64_t)p[2] << 40) | ((uint64_t)p[3] << 32) |
                       ((uint64_t)p[4] << 24) | ((uint64_t)p[5] << 16) | ((uint64_t)p[6] << 8) | (uint64_t)p[7];
    w->win = v << (w->pos & 7);
    w->avail = 64 - (int)(w->pos & 7);
}
static inline uint32_t bw_peek32(BitWin *w) { if (w->avail < 32) bw_refill(w); return (uint32_t)(w->win >> 32); }
static inline void bw_skip(BitWin *w, uint32_t n) { w->win <<= n; w->avail -= (int)n; w->pos += n; }      /* n <= 32, after a peek */
static inline uint32_t bw_get(BitWin *w, uint32_t n) { if (!n) return 0; const uint32_t v = bw_peek32(w) >> (32 - n); bw_skip(w, n); return v; }
static inline int bw_vlc(const Vlc *v, BitWin *w)
{
    const uint32_t x = bw_peek32(w);
    int e = v->l1[x >> 24];
    if (e < 0) e = v->l2[-e - 1][(x >> 16) & 0xFF];
    if (e == VLC_BAD) return -1;
    bw_skip(w, (uint32_t)e >> 8);
    return e & 0xFF;
}
/* hand the position back; 1 = the block ran past the end of the data */
static inline int bw_done(const BitWin *w, BitReader *br)
{
    if (br->pos <= br->size_bits) br->pos = w->pos;
    if (br->pos > br->size_bits) { br->pos = br->size_bits + 1; return 1; }
    return 0;
}

int hd_cavlc_block_sum(BitReader *br, int nc, int max_coeff, int16_t *coef, int *spill, uint32_t *abs_sum)
{
    if (spill) *spill = 0;
    BitWin bw = { br->buf, 0, br->pos, br->size_bits, 0 };
    int sym;
    if (nc < 0) sym = bw_vlc(&vlc_cdc, &bw);
    else if (nc < 2) sym = bw_vlc(&vlc_ct[0], &bw);
    else if (nc < 4) sym = bw_vlc(&vlc_ct[1], &bw);
    else if (nc < 8) sym = bw_vlc(&vlc_ct[2], &bw);
    else {
        /* 6-bit fixed length: 000011 = (0,0), otherwise xxxxyy = (total_coeff-1, trailing_ones) */
        uint32_t c = bw_get(&bw, 6);
        if (c == 3) sym = 0;
        else if ((c & 3) > (c >> 2) + 1) sym = -1;        /* trailing_ones > total_coeff */
        else sym = (int)(((c >> 2) + 1) * 4 + (c & 3));
    }
    if (sym < 0) { bw_done(&bw, br); return -1; }
    const int total = sym >> 2, t1 = sym & 3;
    if (total == 0) { bw_done(&bw, br); return 0; }               /* (an overrun here is the caller's to notice, as before) */
    if (total > max_coeff) { bw_done(&bw, br); return -1; }

    int level[16];
    /* trailing ones */
    if (t1) {
        uint32_t signs = bw_get(&bw, (uint32_t)t1);
        for (int i = 0; i < t1; i++) level[i] = (signs >> (t1 - 1 - i)) & 1 ? -1 : 1;
    }
    /* remaining levels, 9.2.2.1 */
    int suffix_len = (total > 10 && t1 < 3) ? 1 : 0;
    for (int i = t1; i < total; i++) {
        uint32_t w = bw_peek32(&bw);
        if (w < (1u << 16)) { bw_done(&bw, br); return -1; }      /* level_prefix > 15 */
        int prefix = __builtin_clz(w);
        bw_skip(&bw, (uint32_t)prefix + 1);
        int code = prefix << suffix_len;
        int suffix_size = suffix_len;
        if (prefix == 14 && suffix_len == 0) suffix_size = 4;
        else if (prefix == 15) { suffix_size = 12; code = 15 << suffix_len; }
        if (suffix_size) code += (int)bw_get(&bw, (uint32_t)suffix_size);
        if (prefix == 15 && suffix_len == 0) code += 15;
        if (i == t1 && t1 < 3) code += 2;
        level[i] = (code & 1) ? -((code + 1) >> 1) : ((code + 2) >> 1);
        if (suffix_len == 0) suffix_len = 1;
        int mag = level[i] < 0 ? -level[i] : level[i];
        if (mag > (3 << (suffix_len - 1)) && suffix_len < 6) suffix_len++;
    }
    /* total_zeros */
    int zeros_left = 0;
    if (total < max_coeff) {
        zeros_left = nc < 0 ? bw_vlc(&vlc_ctz[total - 1], &bw) : bw_vlc(&vlc_tz[total - 1], &bw);
        /* A 15-coefficient block is parsed with the total_zeros tables of the 16-coefficient case (9.2.3 gives it
         * tzVlcIndex = total_coeff all the same), which allow total_coeff + total_zeros == 16.  The reference accepts
         * that (src/h264bsd_cavlc.c:862-873): the coefficient lands one element past the block, i.e. in element 0 of
         * the NEXT block of residual_t.level[][] (macroblock_layer.c:745-753 pass level[b]+1).  Mirrored: accepted,
         * the stray level is handed to the caller through *spill. */
        if (zeros_left < 0 || total + zeros_left > (max_coeff == 15 ? 16 : max_coeff)) { bw_done(&bw, br); return -1; }
    }
    /* run_before + placement: level[0] is the highest-frequency coefficient */
    int pos = total + zeros_left - 1;                     /* scan index of level[0] */
    const int first = max_coeff == 15 ? 1 : 0;            /* AC blocks start at scan position 1 */
    for (int i = 0; i < total; i++) {
        if (pos < 0) { bw_done(&bw, br); return -1; }
        int16_t v = (int16_t)level[i];
        if (nc < 0) { coef[pos] = v; *abs_sum += (uint32_t)(v < 0 ? -v : v); }
        else if (pos + first < 16) { coef[zigzag4x4[pos + first]] = v; *abs_sum += (uint32_t)(v < 0 ? -v : v); }
        else if (spill) *spill = level[i];
        if (i + 1 == total) break;
        int run = 0;
        if (zeros_left > 0) {
            run = bw_vlc(&vlc_rb[(zeros_left > 7 ? 7 : zeros_left) - 1], &bw);
            if (run < 0 || run > zeros_left) { bw_done(&bw, br); return -1; }
            zeros_left -= run;
        }
        pos -= run + 1;
    }
    if (bw_done(&bw, br)) return -1;
    return total;
}
