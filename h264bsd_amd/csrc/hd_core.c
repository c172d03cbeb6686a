/*
 * hd_core.c — decoder instance life cycle, the per-NAL control flow of h264bsdDecode and the
 * assembly of one frame job per picture.
 *
 * The return-code protocol is the reference's (src/h264bsd_decoder.c:152-515):
 *   - one NAL unit per call, *read_bytes = bytes to advance;
 *   - first slice after a new SPS activation returns HDRS_RDY with read_bytes == 0 and expects to
 *     be called again with the same pointer (:343-389, two-phase activation storage.c:297-419);
 *   - PIC_RDY when the last macroblock of a picture has been parsed (:457-462, :473-510).
 * What is different by design: at PIC_RDY no pixel exists yet.  The picture has been turned into a
 * frame job and queued on the JobSink; pixels materialise when the application pulls the picture.
 * Damaged streams: a corrupt slice is un-decoded (mark_slice_corrupted), and a picture that is still incomplete
 * when the next access unit starts is completed by concealment records (plan_concealment) whose pixels the
 * kernels produce; the call returns PIC_RDY with read_bytes == 0 like the reference (decoder.c:226-266).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "hostdec.h"

HostDec *hd_create(int no_output_reordering);
void hd_destroy(HostDec *d);
/* error exits of hd_decode report where they happened when HD_TRACE is set in the environment (debugging aid) */
#define ERR_RETURN do { if (hd_trace) fprintf(stderr, "TRACE hd_decode error at line %d\n", __LINE__); return HD_ERROR; } while (0)
int hd_decode(HostDec *d, uint8_t *stream, uint32_t len, uint32_t pic_id, uint32_t *read_bytes);

HostDec *hd_create(int no_output_reordering)
{
    HostDec *d = (HostDec *)calloc(1, sizeof(HostDec));
    if (!d) return NULL;
    hd_cavlc_init();
    d->active_sps_id = d->active_pps_id = d->old_sps_id = -1;
    d->no_reordering_app = (uint8_t)(no_output_reordering != 0);
    d->aub_first_call = 1;
    d->dpb.cur = -1;
    return d;
}

void hd_destroy(HostDec *d)
{
    if (!d) return;
    if (d->sink.close) d->sink.close(d->sink.user);
    for (int i = 0; i < HD_MAX_SPS; i++) free(d->sps[i]);
    for (int i = 0; i < HD_MAX_PPS; i++) { hd_free_pps(d->pps[i]); free(d->pps[i]); }
    free(d->mb);
    free(d->mb_decoded);
    free(d->mb_slice_id);
    free(d->mb_rec_sid);
    free(d->slice_group_map);
    free(d->ghost_buf);
    free(d->mb_ghost);
    free(d->redo);
    free(d->mb_redone);
    free(d->redo2);
    free(d->nal_buf);
    if (!d->job_from_sink) free(d->job);
    free(d->conv_buf);
    free(d->tile_ver); free(d->tile_pending);
    free(d);
}

/* ---------------------------------------------------------------- frame job */
/* the parser's dense motion-vector array (FjHeader.mv_off): the tail of the job buffer, never part of the finished job */
static uint32_t job_dense_mv_bytes(uint32_t n_mbs) { return n_mbs * 64u + 32u; }

static uint32_t job_capacity(uint32_t n_mbs)
{
    /* coefficients: 27 blocks per macroblock (16 luma, Intra16x16 DC, 2 chroma DC sharing one, 8 chroma AC) + 2, and room
     * for one more macroblock: a redundant slice that decodes the last macroblock of a full picture again parses its
     * blocks into the section before it gives them back */
    return 128u + n_mbs * 32u + (n_mbs * 27u + 2u + 27u) * 32u
           + (n_mbs + 2u) * 4u /* level starts */ + fj_align32(n_mbs * 2u) /* intra index */
           + n_mbs * 16u /* copy list + general list */ + fj_align32(n_mbs * 2u) /* deblocking index */ + 512u
           + n_mbs * 64u /* sparse vectors, at worst all of them */ + job_dense_mv_bytes(n_mbs);
}

int hd_job_begin(HostDec *d)
{
    const uint32_t n = d->pic_size_mbs, cap = job_capacity(n);
    if (d->sink.acquire) {
        /* build the job in place in the sink's (pinned) staging memory */
        d->job = d->sink.acquire(d->sink.user, cap);
        d->job_from_sink = 1;
        d->job_cap = d->job ? cap : 0;
        if (!d->job) return -1;
    } else if (d->job_cap < cap) {
        free(d->job);
        d->job = (uint8_t *)malloc(cap);
        if (!d->job) { d->job_cap = 0; return -1; }
        d->job_cap = cap;
    }
    FjHeader *h = (FjHeader *)d->job;
    memset(h, 0, sizeof(*h));
    h->magic = FJ_MAGIC;
    h->width_mbs = (uint16_t)d->width_mbs;
    h->height_mbs = (uint16_t)d->height_mbs;
    h->n_mbs = n;
    h->rec_off = 128;
    h->coef_off = h->rec_off + n * 32u;
    h->mv_off = (cap - n * 64u) & ~31u;             /* host only, behind everything the finished job can hold (job_capacity) */
    /* records and motion vectors are NOT pre-initialised: every decoded macroblock writes both, and
     * hd_job_finish() fills in the macroblocks no slice covered (saves two passes over 0.8 MB per 1080p picture) */
    d->coef_blocks = 0;
    d->coef_cap_blocks = n * 27u + 2u + 27u;    /* the share of job_capacity() */
    d->n_inter = d->n_intra = 0;
    d->job_open = 1;
    return 0;
}

/* Derived sections of a frame job whose header (geometry, rec_off, mv_off, coef_off), records, motion vectors
 * and coefficient blocks are filled in: intra dependency levels + schedule, copy runs, general-inter list,
 * non-trivial deblocking index, statistics, total size.  Pure function of those inputs (also exported as
 * h264bsdmiJobFinalize so that tests can hand-craft frame jobs).
 *
 * Dependency level of every intra macroblock: 0 when none of the intra neighbours whose samples its
 * prediction modes actually read (subset of A, B, C, D) precedes it, else 1 + the deepest of them.  Intra MBs of one level are mutually independent, so the device can
 * reconstruct level by level (all inter MBs first). */
static int in_intra_schedule(int kind)
{
    return kind == FJ_MB_I4x4 || kind == FJ_MB_I16x16 || kind == FJ_MB_IPCM || kind == FJ_MB_CONCEAL_I;
}

/* per-thread scratch of fj_finalize (lists are built in one raster pass and copied behind the sections whose sizes
 * are only known afterwards); grows on demand, never shrinks */
static __thread uint8_t *tl_scratch;
static __thread size_t tl_scratch_cap;

int fj_finalize(uint8_t *job, uint32_t cap, uint32_t coef_blocks) { return fj_finalize_ex(job, cap, coef_blocks, NULL); }

int fj_finalize_ex(uint8_t *job, uint32_t cap, uint32_t coef_blocks, FjElide *elide)
{
    FjHeader *h = (FjHeader *)job;
    const uint32_t n = h->n_mbs, w = h->width_mbs;
    FjMbRec *recs = (FjMbRec *)(job + h->rec_off);
    const int16_t (*mvs)[16][2] = (const int16_t (*)[16][2])(job + h->mv_off);
    /* the dense vectors (n x 64 bytes, read from here on) must lie inside the buffer, and whole: in front of the coefficients
     * (hand-built jobs) or behind them; that they also clear the finished job is checked once its size is known */
    if ((size_t)h->mv_off + (size_t)n * 64u > cap) return -1;
    if (h->mv_off < h->coef_off && (size_t)h->mv_off + (size_t)n * 64u > h->coef_off) return -1;
    {
        const size_t need_bytes = (size_t)n * (1 + 8 + 16 + 2 + 2 + 2) + ((size_t)n + 2) * 4 + 64;
        if (tl_scratch_cap < need_bytes) {
            free(tl_scratch);
            tl_scratch = (uint8_t *)malloc(need_bytes);
            tl_scratch_cap = tl_scratch ? need_bytes : 0;
            if (!tl_scratch) return -1;
        }
    }
    /* scratch layout (16-byte entries first for alignment) */
    FjGen *gen_tmp = (FjGen *)tl_scratch;
    FjCopy *copy_tmp = (FjCopy *)(gen_tmp + n);
    uint32_t *hist = (uint32_t *)(copy_tmp + n);                 /* intra MBs per level */
    uint16_t *dbk_tmp = (uint16_t *)(hist + n + 2);
    uint16_t *ilist = dbk_tmp + n;                               /* intra-schedule MBs in raster order */
    uint16_t *mvx_list = ilist + n;                              /* inter macroblocks with more than one vector, raster order */
    uint8_t *cls = (uint8_t *)(mvx_list + n);
    uint32_t max_level = 0, n_intra = 0, n_absent = 0, n_conceal = 0, n_copy = 0, n_gen = 0, n_dbk = 0, n_mvx = 0;
    uint8_t any_dbk = 0;
    memset(hist, 0, ((size_t)n + 2) * 4);

    /* ---- one raster pass: intra dependency levels + FJ_NEED masks, classification of the inter macroblocks, copy
     * runs, general-inter entries, deblocking index ---- */
    const int recon_all = !h->dbk_only;       /* a deblock-only job reconstructs only its FJ_PRED_PHASE2 macroblocks */
#define RECON(r_) (recon_all || ((r_)->pred & FJ_PRED_PHASE2))
    for (uint32_t a = 0; a < n; a++) {
        FjMbRec *r = &recs[a];
        const int recon = RECON(r);
        any_dbk |= r->dbk;
        /* seven of ten macroblocks of a P picture: one vector (the parser's hint), one reference, no coefficients — nothing of
         * the general path below applies to them but the classification and, off the whole-sample grid, a list entry */
        if (r->kind == FJ_MB_INTER && (r->pred & FJ_PRED_UNIFORM_MV) && r->coded == 0) {
            const int whole = ((r->mv[0] | r->mv[1]) & 7) == 0;
            cls[a] = whole ? 3 : 1;
            r->mvx = 0;
            if (recon && !whole) {
                FjGen *gi = &gen_tmp[n_gen++];
                gi->mb = (uint16_t)a; gi->uniform = 1; gi->slot = r->ref_slot[0];
                gi->mvx = r->mv[0]; gi->mvy = r->mv[1]; gi->coef_idx = FJ_GEN_COEF(r->coef_idx, r->qp_y, r->qp_c); gi->coded = 0;
            }
            goto deblock_index;
        }
        cls[a] = 0;
        /* invariant of the job format, checked where the kernels' lists are made: a macroblock that will be reconstructed
         * has its coefficient blocks inside the section (a parser bug must end in a failed decode, not in a kernel
         * reading past the job) */
        if (recon && (r->kind == FJ_MB_INTER || r->kind == FJ_MB_I4x4 || r->kind == FJ_MB_I16x16 || r->kind == FJ_MB_IPCM)) {
            const uint32_t cm = r->coded & 0x03FFFFFFu;         /* (most macroblocks have none: no population count for them) */
            const uint32_t nb = r->kind == FJ_MB_IPCM ? 12u : cm ? (uint32_t)__builtin_popcount(cm) : 0u;
            if (nb && (r->coef_idx > coef_blocks || nb > coef_blocks - r->coef_idx)) {
                if (hd_trace) fprintf(stderr, "TRACE fj_finalize: mb %u kind %u pred %#x needs blocks %u..%u of %u (ghost %u dbk_only %u)\n", a, r->kind, r->pred, r->coef_idx, r->coef_idx + nb, coef_blocks, h->ghost, h->dbk_only);
                return -1;
            }
        }
        if (r->kind == FJ_MB_ABSENT || r->kind == FJ_MB_STALE) n_absent++;
        else if (r->kind == FJ_MB_CONCEAL_I) n_conceal += (uint32_t)recon;
        else if (in_intra_schedule(r->kind)) {
            if (!recon) goto deblock_index;
            const uint32_t x = a % w, y = a / w;
            int lvl = -1;
            /* which neighbouring macroblocks does the prediction of this one actually read? (8.3.1.2, 8.3.3, 8.3.4) */
            unsigned need = 0;
            if (r->kind == FJ_MB_I16x16) {
                const int m = r->pred & 3;
                need |= m == 0 ? FJ_AVAIL_B : m == 1 ? FJ_AVAIL_A : m == 2 ? (FJ_AVAIL_A | FJ_AVAIL_B) : (FJ_AVAIL_A | FJ_AVAIL_B | FJ_AVAIL_D);
            } else if (r->kind == FJ_MB_I4x4) {
                for (int z = 0; z < 16; z++) {
                    const int bx = ((z >> 2) & 1) * 2 + (z & 1), by = (z >> 3) * 2 + ((z >> 1) & 1);
                    if (bx && by) continue;
                    const int m = (r->i4mode[z >> 1] >> ((z & 1) * 4)) & 15;
                    const int uses_left = m == 1 || m == 2 || m == 4 || m == 5 || m == 6 || m == 8;
                    const int uses_top = m == 0 || m == 2 || m == 3 || m == 4 || m == 5 || m == 6 || m == 7;
                    const int uses_corner = m == 4 || m == 5 || m == 6;
                    if (bx == 0 && (uses_left || uses_corner)) need |= FJ_AVAIL_A;
                    if (by == 0 && (uses_top || uses_corner)) need |= FJ_AVAIL_B;
                    if (bx == 0 && by == 0 && uses_corner) need |= FJ_AVAIL_D;
                    if (bx == 3 && by == 0 && (m == 3 || m == 7)) need |= FJ_AVAIL_C;
                }
            }
            if (r->kind != FJ_MB_IPCM) {
                const int m = (r->pred >> 2) & 3;
                need |= m == 0 ? (FJ_AVAIL_A | FJ_AVAIL_B) : m == 1 ? FJ_AVAIL_A : m == 2 ? FJ_AVAIL_B : (FJ_AVAIL_A | FJ_AVAIL_B | FJ_AVAIL_D);
            }
            need &= r->avail;             /* an unavailable neighbour is never read (its samples are replaced by 128) */
#define DEP(cond, idx) do { if (cond) { const FjMbRec *q = &recs[idx]; \
            if (in_intra_schedule(q->kind) && RECON(q) && (int)q->intra_level > lvl) lvl = q->intra_level; } } while (0)
            DEP(x > 0 && (need & FJ_AVAIL_A), a - 1);
            DEP(y > 0 && (need & FJ_AVAIL_B), a - w);
            DEP(y > 0 && x + 1 < w && (need & FJ_AVAIL_C), a - w + 1);
            DEP(y > 0 && x > 0 && (need & FJ_AVAIL_D), a - w - 1);
#undef DEP
            r->intra_level = (uint16_t)(lvl + 1);
            if ((uint32_t)(lvl + 1) > max_level) max_level = (uint32_t)(lvl + 1);
            hist[lvl + 1]++;
            ilist[n_intra++] = (uint16_t)a;
            /* the same information for the device's dataflow scheduler: which of the 8 surrounding macroblocks this
             * one waits for (FJ_NEED_*; the device ignores neighbours that are not in the intra schedule).  Intra
             * macroblocks have no reference slots, so the mask travels in ref_slot[0]. */
            r->ref_slot[0] = (uint8_t)(((need & FJ_AVAIL_A) ? FJ_NEED_L : 0) | ((need & FJ_AVAIL_D) ? FJ_NEED_UL : 0) |
                                       ((need & FJ_AVAIL_B) ? FJ_NEED_U : 0) | ((need & FJ_AVAIL_C) ? FJ_NEED_UR : 0));
            r->ref_slot[1] = r->ref_slot[2] = r->ref_slot[3] = 0;
        } else if (r->kind == FJ_MB_CONCEAL_P) {
            cls[a] = 6;                                            /* copy list, never "uniform" (bit0) */
        } else if (r->kind == FJ_MB_INTER) {
            /* cls bit0: uniform (16 equal mvs, one reference, no coefficients); bit1: additionally whole-sample for
             * luma and chroma -> pure copy */
            /* the parser's hint: one vector, already in the record, the macroblock's dense entry was never written (framejob.h) */
            const int hinted = (r->pred & FJ_PRED_UNIFORM_MV) != 0;
            const int16_t *m0 = hinted ? r->mv : mvs[a][0];
            uint32_t refs;
            memcpy(&refs, r->ref_slot, 4);
            const int one_ref = refs == (refs & 255u) * 0x01010101u;
            int same_mv = one_ref || hinted;
            if (same_mv && !hinted) {                           /* 16 equal vectors: eight 64-bit words equal to the doubled first */
                uint64_t w[8], acc = 0;
                uint32_t first;
                memcpy(w, mvs[a], 64);
                memcpy(&first, m0, 4);
                const uint64_t two = (uint64_t)first << 32 | first;
                for (int k = 0; k < 8; k++) acc |= w[k] ^ two;
                same_mv = acc == 0;
            }
            /* where the finished job keeps the vectors: one in the record, or sixteen in the sparse section (framejob.h) */
            if (!hinted) { r->mv[0] = m0[0]; r->mv[1] = m0[1]; }
            if (same_mv) { r->pred |= FJ_PRED_UNIFORM_MV; r->mvx = 0; }
            else { r->pred &= (uint8_t)~FJ_PRED_UNIFORM_MV; r->mvx = n_mvx; mvx_list[n_mvx++] = (uint16_t)a; }
            const int uni = same_mv && r->coded == 0;
            cls[a] = (uint8_t)(uni ? (((m0[0] | m0[1]) & 7) == 0 ? 3 : 1) : 0);
            if (recon && !(cls[a] & 2)) {
                FjGen *gi = &gen_tmp[n_gen++];
                int quad = 0;
                if (!same_mv) {                               /* one motion vector per 8x8 quadrant (16x8, 8x16, 8x8 partitions)? */
                    quad = 1;
                    for (int q = 0; q < 4 && quad; q++) {
                        const int b0 = (q >> 1) * 8 + (q & 1) * 2;
                        const int16_t *m = mvs[a][b0];
                        quad = mvs[a][b0 + 1][0] == m[0] && mvs[a][b0 + 1][1] == m[1] && mvs[a][b0 + 4][0] == m[0] &&
                               mvs[a][b0 + 4][1] == m[1] && mvs[a][b0 + 5][0] == m[0] && mvs[a][b0 + 5][1] == m[1];
                    }
                }
                gi->mb = (uint16_t)a; gi->uniform = (uint8_t)(same_mv ? 1 : quad ? 2 : 0); gi->slot = r->ref_slot[0];
                gi->mvx = m0[0]; gi->mvy = m0[1]; gi->coef_idx = FJ_GEN_COEF(r->coef_idx, r->qp_y, r->qp_c); gi->coded = r->coded;
                if (!same_mv) memcpy(&gi->mvx, &r->mvx, 4);      /* partitioned: the two fields hold the index of its sixteen vectors */
            }
        }
deblock_index:
        {
            /* deblocking: a uniform MB whose filtered left/top neighbours are uniform too, with the same reference and
             * mv components closer than 4 quarter samples, has all-zero strengths (8.7.2.1) — never visited again */
            int trivial = r->kind == FJ_MB_ABSENT || r->dbk == 0;
            if (!trivial && (cls[a] & 1)) {
                trivial = 1;
                const uint32_t nb[2] = { a - 1, a - w };
                const uint8_t want[2] = { (uint8_t)(r->dbk & FJ_DBK_LEFT), (uint8_t)(r->dbk & FJ_DBK_TOP) };
                for (int k = 0; k < 2 && trivial; k++) {
                    if (!want[k]) continue;
                    const uint32_t p = nb[k];
                    if (cls[p] & 1) {
                        const int dx = r->mv[0] - recs[p].mv[0], dy = r->mv[1] - recs[p].mv[1];
                        trivial = recs[p].ref_slot[0] == r->ref_slot[0] && dx > -4 && dx < 4 && dy > -4 && dy < 4;
                    } else if (recs[p].kind == FJ_MB_INTER) {
                        /* a coded or partitioned inter neighbour: the four 4x4 blocks of it that touch the edge decide (8.7.2.1: a
                         * coded block, another reference or a vector 4 quarter samples away give a strength).  A fifth of what
                         * k_dbk used to visit turned out strength-free: most of it is this case */
                        for (int i = 0; i < 4 && trivial; i++) {
                            const int bx = k ? i : 3, by = k ? 3 : i;
                            const int z = ((by >> 1) << 3) | ((bx >> 1) << 2) | ((by & 1) << 1) | (bx & 1);
                            const int16_t *pm = (recs[p].pred & FJ_PRED_UNIFORM_MV) ? recs[p].mv : mvs[p][4 * by + bx];
                            const int dx = r->mv[0] - pm[0], dy = r->mv[1] - pm[1];
                            trivial = !((recs[p].coded >> z) & 1u) && recs[p].ref_slot[(by >> 1) * 2 + (bx >> 1)] == r->ref_slot[0] &&
                                      dx > -4 && dx < 4 && dy > -4 && dy < 4;
                        }
                    } else trivial = 0;
                }
            }
            r->dbk_trivial = (uint8_t)trivial;
            if (!trivial) dbk_tmp[n_dbk++] = (uint16_t)a;
        }
    }
    /* ---- copy list: runs of up to FJ_COPY_RUN copy MBs with consecutive addresses, equal reference and mv; a run whose
     * displacement is zero may continue into the next row (tiles are contiguous in address order).  Built after the raster
     * pass because copy elision looks at the deblocking verdict of the macroblocks to the right and below: a zero-motion
     * copy is CLEAN when neither its own filter nor that of the two macroblocks that could reach into its tile runs; a
     * clean copy makes the destination tile equal to the source tile, so its tile inherits the source's number, and when
     * the destination carried that number already the copy would write what is there — it is left out. ---- */
    if (elide && !recon_all) elide = NULL;                 /* (callers do not ask for it either) */
    for (uint32_t a = 0, x = 0; a < n; a++, x = x + 1 == w ? 0 : x + 1) {     /* x = a % w without the division */
        const FjMbRec *r = &recs[a];
        if (elide) elide->out[a] = elide->serial;
        if (!(RECON(r) && (cls[a] & 2))) continue;
        const int16_t *m0 = r->mv;
        if (elide && (m0[0] | m0[1]) == 0 && r->ref_slot[0] < elide->n_slots && r->ref_slot[0] != elide->cur_slot && r->dbk_trivial &&
            (x + 1 == w || recs[a + 1].dbk_trivial) && (a + w >= n || recs[a + w].dbk_trivial)) {
            const uint32_t v = elide->ver[(size_t)r->ref_slot[0] * n + a];
            elide->out[a] = v;
            if (elide->ver[(size_t)elide->cur_slot * n + a] == v) { elide->n_elided++; continue; }
        }
        FjCopy *last = n_copy ? &copy_tmp[n_copy - 1] : NULL;
        if (last && last->count < FJ_COPY_RUN && (uint32_t)last->mb + last->count == a && (x != 0 || (m0[0] | m0[1]) == 0) &&
            last->slot == r->ref_slot[0] && last->dx == (m0[0] >> 2) && last->dy == (m0[1] >> 2)) {
            last->count++;
        } else {
            copy_tmp[n_copy].mb = (uint16_t)a; copy_tmp[n_copy].slot = r->ref_slot[0]; copy_tmp[n_copy].count = 1;
            copy_tmp[n_copy].dx = (int16_t)(m0[0] >> 2); copy_tmp[n_copy].dy = (int16_t)(m0[1] >> 2);
            n_copy++;
        }
    }
    if (n_conceal) {
        /* Synthesised macroblocks read the neighbours that were decoded or concealed BEFORE them in the reference's
         * concealment order (coef_idx), on any of the four sides: walk them in that order. */
        uint32_t *ord = (uint32_t *)malloc(n_conceal * sizeof(uint32_t));
        if (!ord) return -1;
        uint32_t k = 0;
        for (uint32_t a = 0; a < n; a++) if (recs[a].kind == FJ_MB_CONCEAL_I && RECON(&recs[a])) ord[k++] = a;
        for (uint32_t i = 1; i < k; i++) {              /* insertion sort by sequence number */
            const uint32_t v = ord[i];
            uint32_t j = i;
            while (j > 0 && recs[ord[j - 1]].coef_idx > recs[v].coef_idx) { ord[j] = ord[j - 1]; j--; }
            ord[j] = v;
        }
        for (uint32_t i = 0; i < k; i++) {
            const uint32_t a = ord[i], x = a % w;
            FjMbRec *r = &recs[a];
            int lvl = -1;
#define DEP(cond, idx) do { if (cond) { const FjMbRec *q = &recs[idx]; \
            if (in_intra_schedule(q->kind) && RECON(q) && (int)q->intra_level > lvl) lvl = q->intra_level; } } while (0)
            DEP((r->avail & FJ_CONC_LEFT) && x > 0, a - 1);
            DEP((r->avail & FJ_CONC_RIGHT) && x + 1 < w, a + 1);
            DEP((r->avail & FJ_CONC_ABOVE) && a >= w, a - w);
            DEP((r->avail & FJ_CONC_BELOW) && a + w < n, a + w);
#undef DEP
            r->intra_level = (uint16_t)(lvl + 1);
            if ((uint32_t)(lvl + 1) > max_level) max_level = (uint32_t)(lvl + 1);
            hist[lvl + 1]++;
            n_intra++;
            r->ref_slot[0] = (uint8_t)(((r->avail & FJ_CONC_LEFT) ? FJ_NEED_L : 0) | ((r->avail & FJ_CONC_ABOVE) ? FJ_NEED_U : 0) |
                                       ((r->avail & FJ_CONC_RIGHT) ? FJ_NEED_R : 0) | ((r->avail & FJ_CONC_BELOW) ? FJ_NEED_D : 0));
            r->ref_slot[1] = r->ref_slot[2] = r->ref_slot[3] = 0;
        }
        free(ord);
    }
    /* ---- sections behind the coefficients: level starts | intra index | copy runs | general-inter | deblocking index */
    const uint32_t n_levels = n_intra ? max_level + 1 : 0;
    h->n_coef_blocks = coef_blocks;
    h->lvl_off = fj_align32(h->coef_off + coef_blocks * 32u);
    h->idx_off = fj_align32(h->lvl_off + (n_levels + 1) * 4u);
    h->copy_off = fj_align32(h->idx_off + n_intra * 2u);
    h->n_copy = n_copy;
    h->n_gen = n_gen;
    h->gen_off = fj_align32(h->copy_off + n_copy * 8u);
    h->dbk_off = fj_align32(h->gen_off + n_gen * 16u);
    h->n_dbk = n_dbk;
    h->mvx_off = fj_align32(h->dbk_off + n_dbk * 2u);
    h->n_mvx = n_mvx;
    h->total_bytes = fj_align32(h->mvx_off + n_mvx * 64u);
    if (h->total_bytes > cap) return -1;
    /* the dense vectors are read below: they lie in front of the coefficients (hand-built jobs) or behind everything (the parser) */
    if (h->mv_off >= h->coef_off && h->mv_off < h->total_bytes) return -1;       /* (behind the coefficients = behind the whole job) */
    for (uint32_t i = 0; i < n_mvx; i++) memcpy(job + h->mvx_off + (size_t)i * 64u, mvs[mvx_list[i]], 64);
    memcpy(job + h->copy_off, copy_tmp, (size_t)n_copy * 8u);
    {   /* general-inter list: the entries with one motion vector per macroblock first (k_recon_inter<0>), then those with
         * one per 8x8 quadrant (<1>), then the finer partitions (<2>); every part keeps raster order */
        FjGen *dst = (FjGen *)(job + h->gen_off);
        uint32_t k = 0;
        for (uint32_t i = 0; i < n_gen; i++) if (gen_tmp[i].uniform == 1) dst[k++] = gen_tmp[i];
        h->n_gen_uniform = k;
        for (uint32_t i = 0; i < n_gen; i++) if (gen_tmp[i].uniform == 2) dst[k++] = gen_tmp[i];
        h->n_gen_quad = k - h->n_gen_uniform;
        for (uint32_t i = 0; i < n_gen; i++) if (gen_tmp[i].uniform == 0) dst[k++] = gen_tmp[i];
    }
    memcpy(job + h->dbk_off, dbk_tmp, (size_t)n_dbk * 2u);
    {   /* intra schedule: the scheduled MB addresses sorted by dependency level, ascending address inside a level */
        uint32_t *lvl_start = (uint32_t *)(job + h->lvl_off);
        uint16_t *idx = (uint16_t *)(job + h->idx_off);
        lvl_start[0] = 0;
        for (uint32_t l = 0; l < n_levels; l++) lvl_start[l + 1] = lvl_start[l] + hist[l];
        for (uint32_t l = 0; l < n_levels; l++) hist[l] = lvl_start[l];           /* hist becomes the cursor */
        if (!n_conceal) {
            for (uint32_t i = 0; i < n_intra; i++) idx[hist[recs[ilist[i]].intra_level]++] = ilist[i];
        } else {
            for (uint32_t a = 0; a < n; a++)
                if (in_intra_schedule(recs[a].kind) && RECON(&recs[a])) idx[hist[recs[a].intra_level]++] = (uint16_t)a;
        }
    }
    {   /* zero the alignment gaps so that a frame job is a pure function of the bitstream */
        const uint32_t ends[6] = { h->coef_off + coef_blocks * 32u, h->lvl_off + (n_levels + 1) * 4u,
                                   h->idx_off + n_intra * 2u, h->copy_off + h->n_copy * 8u, h->gen_off + h->n_gen * 16u,
                                   h->dbk_off + h->n_dbk * 2u };
        const uint32_t nexts[6] = { h->lvl_off, h->idx_off, h->copy_off, h->gen_off, h->dbk_off, h->mvx_off };
        for (int i = 0; i < 6; i++) if (nexts[i] > ends[i]) memset(job + ends[i], 0, nexts[i] - ends[i]);
    }
    h->n_intra = n_intra;
    h->n_intra_levels = n_levels;
    h->intra_down_deps = n_conceal ? 1 : 0;      /* concealed macroblocks may wait for the macroblock below them (FJ_NEED_D) */
    h->n_inter = n - n_intra - n_absent;
    h->any_deblock = any_dbk ? 1 : 0;
    return 0;
}

/* safety net: after concealment every macroblock is decoded, but never leave uninitialised records */
static void fill_undecoded(HostDec *d)
{
    FjHeader *h = (FjHeader *)d->job;
    /* The reference's macroblock counter can say "complete" while a macroblock was never decoded (redundant slices and
     * roll-backs miscount, storage.c:538): such a macroblock — touched by a slice that failed before it wrote a record, or
     * by none — has no record, and the job buffer is reused from picture to picture.  (A macroblock that a failed
     * redundant slice un-decoded keeps the record, and the pixels, of the earlier slice: mb_rec_sid.) */
    const int complete = d->num_decoded_mbs == d->pic_size_mbs;
    if (complete && !d->pic_irregular) return;
    FjMbRec *recs = (FjMbRec *)(d->job + h->rec_off);
    const uint32_t w = d->width_mbs;
    for (uint32_t a = 0; a < d->pic_size_mbs; a++)
        if (!d->mb_decoded[a] && !(complete && d->mb_rec_sid[a])) {
            memset(&recs[a], 0, sizeof(FjMbRec));
            recs[a].kind = FJ_MB_ABSENT;
            if (!complete) continue;                   /* (an incomplete picture is concealed: every macroblock gets a record) */
            /* Counter-complete picture, macroblock never decoded: nothing writes its pixels, but h264bsdFilterPicture filters
             * every macroblock with what its mbStorage_t holds (src/h264bsd_deblocking.c:236-275 and :575-640) — type, QP,
             * coefficient counts and motion of the LAST decode of this macroblock in any earlier picture, slice-level
             * parameters of the last slice that started on it (MbInfo persists like mbStorage_t).  The picture becomes the
             * two-job form of redo_split: no record in the reconstruction job, this record in the deblock-only job. */
            const MbInfo *m = &d->mb[a];
            FjMbRec absent = recs[a], keep;
            int16_t zero_mv[32] = { 0 };
            if (hd_redo_keep_first(d, a, &absent, zero_mv)) continue;         /* out of memory: stays absent */
            memset(&keep, 0, sizeof(keep));
            const int intra = m->mb_type >= 6;
            keep.kind = intra ? FJ_MB_STALE : FJ_MB_INTER;                  /* (FJ_MB_STALE: filtered as intra, pixels untouched in any job) */
            keep.qp_y = m->qp;
            keep.alpha_off = m->alpha_off; keep.beta_off = m->beta_off; keep.cqp_off = m->cqp_off;
            if (m->dbk_idc != 1) {
                const int same_l = m->dbk_idc != 2 || (a % w && d->mb_slice_id[a - 1] == d->mb_slice_id[a]);
                const int same_t = m->dbk_idc != 2 || (a >= w && d->mb_slice_id[a - w] == d->mb_slice_id[a]);
                keep.dbk = (uint8_t)(FJ_DBK_INNER | ((a % w) && same_l ? FJ_DBK_LEFT : 0) | (a >= w && same_t ? FJ_DBK_TOP : 0));
            }
            if (!intra) {
                for (int z = 0; z < 16; z++) if (m->tc[z]) keep.coded |= 1u << z;
                const int parts = m->mb_type <= 1 ? FJ_PARTS_16x16 : m->mb_type == 2 ? FJ_PARTS_16x8 : m->mb_type == 3 ? FJ_PARTS_8x16 : FJ_PARTS_8x8;
                keep.pred = (uint8_t)(parts << FJ_PRED_PARTS_SHIFT);
                memcpy(keep.ref_slot, m->ref_slot, 4);
                int16_t (*dst)[2] = (int16_t (*)[2])(d->job + h->mv_off + (size_t)a * 64u);
                static const uint8_t zx[16] = { 0, 1, 0, 1, 2, 3, 2, 3, 0, 1, 0, 1, 2, 3, 2, 3 }, zy[16] = { 0, 0, 1, 1, 0, 0, 1, 1, 2, 2, 3, 3, 2, 2, 3, 3 };
                for (int z = 0; z < 16; z++) { dst[4 * zy[z] + zx[z]][0] = m->mv[z][0]; dst[4 * zy[z] + zx[z]][1] = m->mv[z][1]; }
            }
            recs[a] = keep;
            d->mb_rec_sid[a] = d->mb_slice_id[a] ? d->mb_slice_id[a] : 0xFFFFFFFFu;      /* the record is settled (hd_job_finish calls this again) */
        }
}

/* ---- copy elision: the numbers of the tiles (hostdec.h) ---- */
static void tiles_fresh_slot(HostDec *d, uint32_t slot)
{
    const uint32_t v = ++d->tile_serial;
    uint32_t *t = d->tile_ver + (size_t)slot * d->tile_mbs;
    for (uint32_t a = 0; a < d->tile_mbs; a++) t[a] = v;
}

static void tiles_forget(HostDec *d)
{
    for (uint32_t s = 0; d->tile_ver && s < d->tile_slots; s++) tiles_fresh_slot(d, s);
    d->tile_uncommitted = 0;
}

/* a new sequence: the sink (re)allocates the frames */
static int tiles_reset(HostDec *d, uint32_t n_slots, uint32_t n_mbs)
{
    free(d->tile_ver); free(d->tile_pending);
    d->tile_ver = (uint32_t *)malloc((size_t)n_slots * n_mbs * sizeof(uint32_t));
    d->tile_pending = (uint32_t *)malloc((size_t)n_mbs * sizeof(uint32_t));
    d->tile_slots = n_slots; d->tile_mbs = n_mbs;
    if (!d->tile_ver || !d->tile_pending) { free(d->tile_ver); free(d->tile_pending); d->tile_ver = d->tile_pending = NULL; d->tile_slots = 0; return -1; }
    tiles_forget(d);
    return 0;
}

/* the sink has taken the picture's last job: its tiles are what the job said */
static void tiles_commit(HostDec *d)
{
    if (!d->tile_uncommitted) return;
    memcpy(d->tile_ver + (size_t)d->tile_pending_slot * d->tile_mbs, d->tile_pending, (size_t)d->tile_mbs * sizeof(uint32_t));
    d->tile_uncommitted = 0;
}

/* single_job: the picture is this one job (no reconstruction-only jobs before it, not a deblock-only job): the only case in
 * which copies are left out; any other picture just gives all its tiles a new number */
int hd_job_finish(HostDec *d, int is_idr, int single_job)
{
    FjHeader *h = (FjHeader *)d->job;
    fill_undecoded(d);
    const uint32_t cur = (uint32_t)hd_dpb_cur_slot(&d->dpb);
    FjElide el, *elide = NULL;
    d->n_elided = 0;
    if (d->tile_ver && d->tile_mbs == h->n_mbs && cur < d->tile_slots) {
        if (d->tile_uncommitted) tiles_forget(d);             /* the previous picture never reached the sink */
        /* submit() succeeding only means "queued": if the device has since reported an error (a tripwire of the kernels, a
         * scheduler that gave up), some picture was not produced as its job said — nothing the buffers hold is relied on any
         * more, for the rest of this decoder's life */
        if (d->copy_elision && d->sink.errors && d->sink.errors(d->sink.user) != d->sink_errors_at_start) {
            /* (errors that the device reported before this decoder existed — another decoder's, a test's hand-built job — say
             * nothing about ITS frame buffers: only events since then count) */
            fprintf(stderr, "h264bsd-mi355x: the device reported an error since this decoder was created: copy elision is off for the rest of its life\n");
            d->copy_elision = 0; tiles_forget(d);
        }
        if (d->tile_serial > 0xFFFF0000u) { d->tile_serial = 0; tiles_forget(d); }    /* (numbers are compared for equality: no wrap-around) */
        /* An IDR picture starts a sequence that must be decodable on its own — replay sets start there — so nothing that
         * the other slots held before it is relied on afterwards */
        if (is_idr) for (uint32_t s = 0; s < d->tile_slots; s++) if (s != cur) tiles_fresh_slot(d, s);
        el.ver = d->tile_ver; el.n_slots = d->tile_slots; el.cur_slot = cur; el.serial = ++d->tile_serial;
        el.out = d->tile_pending; el.n_elided = 0;
        /* (the reconstruction-only job that conceals a whole lost picture into the spare buffer, hd_decode: frame_num gap, is a
         * single job too and is elided like any other: the engine executes it like any other, and its tiles are committed) */
        if (d->copy_elision && single_job && !h->dbk_only) elide = &el;
        else for (uint32_t a = 0; a < d->tile_mbs; a++) d->tile_pending[a] = el.serial;
        d->tile_pending_slot = cur;
        d->tile_uncommitted = 1;
    }
    if (fj_finalize_ex(d->job, d->job_cap, d->coef_blocks, elide)) return -1;
    if (elide) d->n_elided = el.n_elided;
    h->cur_slot = (uint8_t)cur;
    h->n_slots = (uint8_t)d->dpb.n_slots;
    h->is_idr = (uint8_t)is_idr;
    h->pic_seq = d->pic_seq++;
    d->job_open = 0;
    return 0;
}

/* ---------------------------------------------------------------- parameter-set activation */
/* reference CheckPps, src/h264bsd_storage.c:795-860: slice-group parameters against the picture size of the SPS */
int hd_check_pps(const Pps *p, const Sps *s)
{
    const uint32_t n = s->width_mbs * s->height_mbs;
    if (p->num_slice_groups > 1) {
        if (p->slice_group_map_type == 0) {
            for (uint32_t i = 0; i < p->num_slice_groups; i++) if (p->run_length[i] > n) return -1;
        } else if (p->slice_group_map_type == 2) {
            for (uint32_t i = 0; i + 1 < p->num_slice_groups; i++)
                if (p->top_left[i] > p->bottom_right[i] || p->bottom_right[i] >= n ||
                    (p->top_left[i] % s->width_mbs) > (p->bottom_right[i] % s->width_mbs)) return -1;
        } else if (p->slice_group_map_type >= 3 && p->slice_group_map_type <= 5) {
            if (p->slice_group_change_rate > n) return -1;
        } else if (p->slice_group_map_type == 6 && p->pic_size_in_map_units < n) return -1;
    }
    return 0;
}

static void select_sps(HostDec *d, int pps_id)
{
    d->active_pps_id = pps_id;
    d->active_pps = d->pps[pps_id];
    d->active_sps_id = d->active_pps->sps_id;
    d->active_sps = d->sps[d->active_sps_id];
    d->width_mbs = d->active_sps->width_mbs;
    d->height_mbs = d->active_sps->height_mbs;
    d->pic_size_mbs = d->width_mbs * d->height_mbs;
    d->width_magic = d->pic_size_mbs < 65536u && d->width_mbs > 1u ? (uint32_t)(0x100000000ull / d->width_mbs) + 1u : 0u;     /* (width 1: 2^32 + 1 does not fit) */
    d->pending_activation = 1;
}

/* returns 0, -1 (bad parameter sets) or -2 (out of memory / engine failure) */
static int activate_param_sets(HostDec *d, uint32_t pps_id, int is_idr)
{
    Pps *p = d->pps[pps_id];
    if (!p || !d->sps[p->sps_id]) return -1;
    if (hd_check_pps(p, d->sps[p->sps_id])) return -1;

    if (d->active_pps_id < 0) {
        /* nothing active yet: phase 1 */
        select_sps(d, (int)pps_id);
    } else if (d->pending_activation) {
        /* phase 2: the caller has seen HDRS_RDY; allocate per-sequence state */
        d->pending_activation = 0;
        free(d->mb);
        free(d->slice_group_map);
        d->mb = (MbInfo *)calloc(d->pic_size_mbs, sizeof(MbInfo));
        free(d->mb_decoded);
        free(d->mb_slice_id);
        free(d->mb_rec_sid);
        free(d->mb_ghost); d->mb_ghost = NULL; d->ghost_dirty = 0; d->ghost_len = 0;
        free(d->mb_redone); d->mb_redone = NULL; d->n_redo = 0; d->n_redo2 = 0;
        d->mb_decoded = (uint8_t *)calloc(d->pic_size_mbs, 1);
        d->mb_slice_id = (uint32_t *)calloc(d->pic_size_mbs, sizeof(uint32_t));
        d->mb_rec_sid = (uint32_t *)calloc(d->pic_size_mbs, sizeof(uint32_t));
        d->slice_group_map = (uint32_t *)calloc(d->pic_size_mbs, sizeof(uint32_t));
        if (!d->mb || !d->slice_group_map || !d->mb_decoded || !d->mb_slice_id || !d->mb_rec_sid) return -2;
        for (uint32_t i = 0; i < d->pic_size_mbs; i++) { d->mb[i].kind = FJ_MB_ABSENT; memset(d->mb[i].ref_slot, 0xFF, 4); }   /* 0xFF: refAddr NULL */
        const Sps *s = d->active_sps;
        int no_reorder = d->no_reordering_app || s->poc_type == 2 ||
                         (s->vui_present && s->bitstream_restriction && s->num_reorder_frames == 0);
        if (hd_dpb_reset(&d->dpb, s->max_dpb_size, s->num_ref_frames, s->max_frame_num, no_reorder)) return -1;
        if (d->sink.configure &&
            d->sink.configure(d->sink.user, d->width_mbs, d->height_mbs, d->dpb.n_slots)) return -2;
        if (tiles_reset(d, d->dpb.n_slots, d->pic_size_mbs)) return -2;
        d->sink_configured = 1;
    } else if ((int)pps_id != d->active_pps_id) {
        if (p->sps_id != d->active_sps_id) {
            if (!is_idr) return -1;          /* the SPS may only change at an IDR picture */
            select_sps(d, (int)pps_id);
        } else {
            d->active_pps_id = (int)pps_id;
            d->active_pps = p;
        }
    }
    return 0;
}

/* ---------------------------------------------------------------- access-unit boundary, 7.4.1.2.4 */
static int check_access_unit_boundary(HostDec *d, const BitReader *br0, int nal_type, int nal_ref_idc, int *boundary)
{
    *boundary = 0;
    if ((nal_type > 5 && nal_type < 12) || (nal_type > 12 && nal_type <= 18)) { *boundary = 1; return 0; }
    if (nal_type != 1 && nal_type != 5) return 0;
    if (d->aub_first_call) { *boundary = 1; d->aub_first_call = 0; }

    uint32_t pps_id;
    if (hd_peek_pps_id(br0, &pps_id)) return -1;
    const Pps *p = d->pps[pps_id];
    if (!p || !d->sps[p->sps_id] ||
        (d->active_sps_id >= 0 && p->sps_id != d->active_sps_id && nal_type != 5))
        return -2;
    const Sps *s = d->sps[p->sps_id];

    if (d->prev_nal_ref_idc != nal_ref_idc && (d->prev_nal_ref_idc == 0 || nal_ref_idc == 0)) *boundary = 1;
    if ((d->prev_nal_type == 5) != (nal_type == 5)) *boundary = 1;

    /* re-read the leading slice-header fields with the candidate SPS/PPS */
    BitReader br = *br0;
    br_ue(&br); br_ue(&br); br_ue(&br);
    uint32_t nb = 0;
    while ((1u << nb) < s->max_frame_num) nb++;
    uint32_t frame_num = br_get(&br, nb);
    if (br_overrun(&br)) return -1;
    if (d->aub_prev_frame_num != frame_num) { d->aub_prev_frame_num = frame_num; *boundary = 1; }
    if (nal_type == 5) {
        uint32_t idr_pic_id = br_ue(&br);
        if (br_overrun(&br)) return -1;
        if (d->prev_nal_type == 5 && d->aub_prev_idr_pic_id != idr_pic_id) *boundary = 1;
        d->aub_prev_idr_pic_id = idr_pic_id;
    }
    if (s->poc_type == 0) {
        nb = 0;
        while ((1u << nb) < s->max_poc_lsb) nb++;
        uint32_t lsb = br_get(&br, nb);
        if (br_overrun(&br)) return -1;
        if (d->aub_prev_poc_lsb != lsb) { d->aub_prev_poc_lsb = lsb; *boundary = 1; }
        if (p->pic_order_present) {
            int32_t db = br_se(&br);
            if (br_overrun(&br)) return -1;
            if (d->aub_prev_delta_poc_bottom != db) { d->aub_prev_delta_poc_bottom = db; *boundary = 1; }
        }
    } else if (s->poc_type == 1 && !s->delta_pic_order_always_zero) {
        int32_t d0 = br_se(&br), d1 = 0;
        if (p->pic_order_present) d1 = br_se(&br);
        if (br_overrun(&br)) return -1;
        if (d->aub_prev_delta_poc[0] != d0) { d->aub_prev_delta_poc[0] = d0; *boundary = 1; }
        if (p->pic_order_present && d->aub_prev_delta_poc[1] != d1) { d->aub_prev_delta_poc[1] = d1; *boundary = 1; }
    }
    d->prev_nal_type = (uint8_t)nal_type;
    d->prev_nal_ref_idc = (uint8_t)nal_ref_idc;
    return 0;
}

static int end_of_picture(const HostDec *d)
{
    if (!d->slice.redundant_pic_cnt) return d->num_decoded_mbs == d->pic_size_mbs;
    uint32_t n = 0;
    for (uint32_t i = 0; i < d->pic_size_mbs; i++) n += d->mb_decoded[i] != 0;
    return n == d->pic_size_mbs;
}

static void reset_picture_state(HostDec *d)
{
    d->num_decoded_mbs = 0;
    d->slice_id = 0;
    memset(d->mb_decoded, 0, d->pic_size_mbs);
    memset(d->mb_slice_id, 0, (size_t)d->pic_size_mbs * sizeof(uint32_t));
    memset(d->mb_rec_sid, 0, (size_t)d->pic_size_mbs * sizeof(uint32_t));
    if (d->ghost_dirty && d->mb_ghost) memset(d->mb_ghost, 0, d->pic_size_mbs);
    d->ghost_dirty = d->ghost_needed = 0;
    d->ghost_len = 0;
    if (d->n_redo && d->mb_redone) memset(d->mb_redone, 0, d->pic_size_mbs);
    d->n_redo = 0; d->n_redo2 = 0;
    d->pic_irregular = 0;
    d->slice_ids_rewritten = 0;
}

/* ---------------------------------------------------------------- parameter set storage */
static int store_sps(HostDec *d, const Sps *s)
{
    const int id = s->sps_id;
    if (!d->sps[id]) {
        d->sps[id] = (Sps *)malloc(sizeof(Sps));
        if (!d->sps[id]) return -2;
    } else if (id == d->active_sps_id) {
        if (hd_sps_equal(s, d->active_sps)) return 0;      /* identical re-send: keep everything */
        d->active_sps_id = HD_MAX_SPS + 1;                  /* changed: force re-activation      */
        d->active_pps_id = HD_MAX_PPS + 1;
        d->active_sps = NULL;
        d->active_pps = NULL;
    }
    *d->sps[id] = *s;
    return 0;
}
static int store_pps(HostDec *d, Pps *p)
{
    const int id = p->pps_id;
    if (!d->pps[id]) {
        d->pps[id] = (Pps *)calloc(1, sizeof(Pps));
        if (!d->pps[id]) { hd_free_pps(p); return -2; }
    } else {
        if (id == d->active_pps_id && p->sps_id != d->active_sps_id) d->active_pps_id = HD_MAX_PPS + 1;
        hd_free_pps(d->pps[id]);
    }
    *d->pps[id] = *p;          /* takes ownership of slice_group_id */
    if (id == d->active_pps_id) d->active_pps = d->pps[id];
    return 0;
}

/* ---------------------------------------------------------------- error handling: lost macroblocks */
/* Ghost pixels.  The reference reconstructs every macroblock while it parses, so when a slice fails and
 * h264bsdMarkSliceCorrupted un-decodes its macroblocks, their PIXELS stay in the frame buffer.  Concealment overwrites
 * them at the end of the access unit — except where a later slice leaves a macroblock "decoded" without ever writing it
 * (FJ_MB_STALE, hd_mb.c): that macroblock shows what the rolled-back slice had put there.  Here pixels are made from
 * the finished frame job, so the rolled-back slice has to be kept: every macroblock it had decoded (the ones that
 * stay decoded too: intra prediction inside the slice reads them) is copied into a side store, and if at the end of
 * the picture a stale macroblock sits on such pixels the stored slices are submitted as reconstruction-only frame jobs
 * (FjHeader.ghost) in front of the picture's own job, into the same DPB slot.  Costs nothing on intact streams. */
/* The sixteen vectors (raster 4x4 order) a record stands for while a picture is being parsed: an inter macroblock with the
 * parser's FJ_PRED_UNIFORM_MV hint has ONE, in the record, and its entry of the dense array was never written; any other inter
 * macroblock has them in the dense array; everything else has none. */
void hd_dense_mv_read(const FjMbRec *r, const int16_t *dense, int16_t out[32])
{
    if (r->kind != FJ_MB_INTER) memset(out, 0, 64);
    else if (r->pred & FJ_PRED_UNIFORM_MV) for (int b = 0; b < 16; b++) { out[2 * b] = r->mv[0]; out[2 * b + 1] = r->mv[1]; }
    else memcpy(out, dense, 64);
}

static int makes_pixels(int kind)
{
    return kind == FJ_MB_INTER || kind == FJ_MB_I4x4 || kind == FJ_MB_I16x16 || kind == FJ_MB_IPCM;
}
static uint32_t rec_blocks(const FjMbRec *r)
{
    return r->kind == FJ_MB_IPCM ? 12u : (uint32_t)__builtin_popcount(r->coded & 0x03FFFFFFu);
}
typedef struct GhostMb { uint32_t addr, n_blocks; FjMbRec rec; int16_t mv[32]; } GhostMb;   /* + n_blocks * 32 bytes */

/* first decode of a macroblock that a redundant slice decoded again (RedoMb), or NULL */
static const struct RedoMb *redo_first_of(const HostDec *d, uint32_t addr)
{
    if (!d->mb_redone || !d->mb_redone[addr]) return NULL;
    for (uint32_t i = 0; i < d->n_redo; i++) if (d->redo[i].addr == addr) return &d->redo[i];
    return NULL;
}

static void ghost_store_slice(HostDec *d, uint32_t sid)
{
    const FjHeader *h = (const FjHeader *)d->job;
    const FjMbRec *recs = (const FjMbRec *)(d->job + h->rec_off);
    const uint32_t n = d->pic_size_mbs;
    size_t need = sizeof(uint32_t);
    uint32_t count = 0, wrote = 0;
    /* The macroblocks this slice WROTE (first decode), and with them those it only decoded again (a redundant slice:
     * pixels of their first decode stay): the written ones may have predicted from them — same slice id — and the
     * ghost job runs before the job that reconstructs first decodes, so it brings them along in that first version. */
    for (uint32_t a = 0; a < n; a++) {
        if (d->mb_slice_id[a] != sid) continue;
        const int own_rec = d->mb_decoded[a] == 1 && d->mb_rec_sid[a] == sid;      /* (see the second loop) */
        const struct RedoMb *f = !own_rec && d->mb_decoded[a] >= 1 ? redo_first_of(d, a) : NULL;
        const FjMbRec *r = own_rec ? &recs[a] : f ? &f->rec : NULL;
        if (!r || !makes_pixels(r->kind)) continue;
        need += sizeof(GhostMb) + (size_t)rec_blocks(r) * 32u;
        count++;
        wrote += d->mb_decoded[a] == 1;
    }
    if (!wrote) return;
    /* bounded: a hostile stream can repeat a failing slice of skipped macroblocks for a few bytes each */
    if (d->ghost_len + need > 4u * (size_t)job_capacity(n)) return;
    if (!d->mb_ghost) d->mb_ghost = (uint8_t *)calloc(n, 1);
    if (d->ghost_len + need > d->ghost_cap) {
        const size_t cap = (d->ghost_len + need) * 2;
        uint8_t *nb = (uint8_t *)realloc(d->ghost_buf, cap);
        if (!nb) return;                              /* out of memory: the deviation is harmless, the stream is damaged */
        d->ghost_buf = nb; d->ghost_cap = cap;
    }
    if (!d->mb_ghost) return;
    uint8_t *p = d->ghost_buf + d->ghost_len;
    memcpy(p, &count, sizeof(count)); p += sizeof(count);
    for (uint32_t a = 0; a < n; a++) {
        if (d->mb_slice_id[a] != sid) continue;
        /* recs[a] belongs to this slice only while no later (redundant) decode replaced it: after such a decode was rolled
         * back (counter 2 -> 1) the record — and the coefficient blocks it points to — are the discarded decode's */
        const int own_rec = d->mb_decoded[a] == 1 && d->mb_rec_sid[a] == sid;
        const struct RedoMb *f = !own_rec && d->mb_decoded[a] >= 1 ? redo_first_of(d, a) : NULL;
        const FjMbRec *r = own_rec ? &recs[a] : f ? &f->rec : NULL;
        if (!r || !makes_pixels(r->kind)) continue;
        GhostMb g;
        g.addr = a; g.n_blocks = rec_blocks(r); g.rec = *r;
        g.rec.pred &= (uint8_t)~FJ_PRED_PHASE2;
        if (f) memcpy(g.mv, f->mv, 64); else hd_dense_mv_read(r, (const int16_t *)(d->job + h->mv_off + (size_t)a * 64u), g.mv);
        g.rec.pred &= (uint8_t)~FJ_PRED_UNIFORM_MV;       /* (the copy is sixteen vectors: fj_finalize looks again) */
        memcpy(p, &g, sizeof(g)); p += sizeof(g);
        memcpy(p, d->job + h->coef_off + (size_t)r->coef_idx * 32u, (size_t)g.n_blocks * 32u);
        p += (size_t)g.n_blocks * 32u;
    }
    d->ghost_len = (size_t)(p - d->ghost_buf);
}

/* one reconstruction-only frame job per stored slice, in the order in which they were decoded */
static int ghost_submit(HostDec *d)
{
    const FjHeader *mh = (const FjHeader *)d->job;
    const uint32_t n = d->pic_size_mbs, cap = job_capacity(n);
    uint8_t *blob = (uint8_t *)malloc(cap);
    if (!blob) return -1;
    int rc = 0;
    for (const uint8_t *p = d->ghost_buf, *end = d->ghost_buf + d->ghost_len; p < end && !rc;) {
        uint32_t count, blocks = 0;
        memcpy(&count, p, sizeof(count)); p += sizeof(count);
        FjHeader *h = (FjHeader *)blob;
        memset(h, 0, sizeof(*h));
        h->magic = FJ_MAGIC; h->width_mbs = mh->width_mbs; h->height_mbs = mh->height_mbs; h->n_mbs = n;
        h->rec_off = 128; h->mv_off = h->rec_off + n * 32u; h->coef_off = h->mv_off + n * 64u;
        FjMbRec *recs = (FjMbRec *)(blob + h->rec_off);
        memset(recs, 0, (size_t)n * 32u);
        for (uint32_t a = 0; a < n; a++) recs[a].kind = FJ_MB_ABSENT;
        memset(blob + h->mv_off, 0, (size_t)n * 64u);
        for (uint32_t i = 0; i < count; i++) {
            GhostMb g;
            memcpy(&g, p, sizeof(g)); p += sizeof(g);
            g.rec.coef_idx = blocks;
            g.rec.dbk = 0;                           /* deblocking happens once, on the picture's own job */
            recs[g.addr] = g.rec;
            memcpy(blob + h->mv_off + (size_t)g.addr * 64u, g.mv, 64);
            memcpy(blob + h->coef_off + (size_t)blocks * 32u, p, (size_t)g.n_blocks * 32u);
            p += (size_t)g.n_blocks * 32u;
            blocks += g.n_blocks;
        }
        if (fj_finalize(blob, cap, blocks)) { rc = -1; break; }
        h->cur_slot = mh->cur_slot; h->n_slots = mh->n_slots; h->is_idr = mh->is_idr; h->pic_seq = mh->pic_seq;
        h->ghost = 1;
        if (d->sink.submit && d->sink.submit(d->sink.user, blob, h->total_bytes)) rc = -1;
    }
    free(blob);
    return rc;
}

/* ---- macroblocks decoded twice (redundant slices in a damaged picture; hostdec.h, RedoMb) ---- */
/* mv: the macroblock's entry of the dense array (only valid where hd_dense_mv_read says so) */
int hd_redo_keep_first(HostDec *d, uint32_t addr, const FjMbRec *rec, const int16_t *mv)
{
    if (!d->mb_redone && !(d->mb_redone = (uint8_t *)calloc(d->pic_size_mbs, 1))) return -1;
    if (d->mb_redone[addr]) {
        /* decoded a third time: the FIRST decode made the pixels — unless the record being replaced is itself a version
         * that was written over them (FJ_PRED_PHASE2); that one still has to be reconstructed, after the first decodes
         * and before whatever the final records add */
        if (!(rec->pred & FJ_PRED_PHASE2) || !makes_pixels(rec->kind)) return 0;
        struct RedoMb *r = NULL;
        for (uint32_t i = 0; i < d->n_redo2; i++) if (d->redo2[i].addr == addr) r = &d->redo2[i];
        if (!r) {
            if (d->n_redo2 == d->redo2_cap) {
                const uint32_t cap = d->redo2_cap ? 2 * d->redo2_cap : 16;
                struct RedoMb *nb = (struct RedoMb *)realloc(d->redo2, (size_t)cap * sizeof(*nb));
                if (!nb) return -1;
                d->redo2 = nb; d->redo2_cap = cap;
            }
            r = &d->redo2[d->n_redo2++];
        }
        r->addr = addr; r->rec = *rec;
        hd_dense_mv_read(rec, mv, r->mv);
        r->rec.pred &= (uint8_t)~FJ_PRED_UNIFORM_MV;
        return 0;
    }
    if (d->n_redo == d->redo_cap) {
        const uint32_t cap = d->redo_cap ? 2 * d->redo_cap : 64;
        struct RedoMb *nb = (struct RedoMb *)realloc(d->redo, (size_t)cap * sizeof(*nb));
        if (!nb) return -1;
        d->redo = nb; d->redo_cap = cap;
    }
    struct RedoMb *r = &d->redo[d->n_redo++];
    if (hd_trace) fprintf(stderr, "TRACE redo first version: mb %u kind %u coef_idx %u rec_sid %u decoded %u\n", addr, rec->kind, rec->coef_idx, d->mb_rec_sid[addr], d->mb_decoded[addr]);
    r->addr = addr; r->rec = *rec;
    hd_dense_mv_read(rec, mv, r->mv);
    r->rec.pred &= (uint8_t)~FJ_PRED_UNIFORM_MV;      /* (sixteen vectors from here on: fj_finalize looks again) */
    d->mb_redone[addr] = 1;
    return 0;
}

/* End of a picture with such macroblocks: returns a malloc'ed reconstruction-only job — the picture's records with the
 * first decodes put back, no deblocking — and turns the picture's own job into a deblock-only one.  Both still have to
 * be finalized. */
/* A redundant slice restamps the slice id of every macroblock it starts on (slice_data.c:140, before the macroblock is
 * parsed), also when it then fails and is rolled back.  The reference decides "slice boundary" for
 * disable_deblocking_filter_idc 2 when it filters (deblocking.c:236-275), from the ids as they are then: the edge flags
 * that were derived while parsing are derived again from the final ids. */
static void restamp_slice_edges(HostDec *d)
{
    FjHeader *h = (FjHeader *)d->job;
    const uint32_t n = d->pic_size_mbs, w = d->width_mbs;
    FjMbRec *recs = (FjMbRec *)(d->job + h->rec_off);
    for (uint32_t a = 0; a < n; a++) {
        if (!d->mb_rec_sid[a] || !(recs[a].dbk & FJ_DBK_INNER) || d->mb[a].dbk_idc != 2 ||
            recs[a].kind == FJ_MB_CONCEAL_P || recs[a].kind == FJ_MB_CONCEAL_I) continue;     /* (concealed: filtered everywhere, conceal.c:305-313) */
        if (a % w) recs[a].dbk = (uint8_t)((recs[a].dbk & ~FJ_DBK_LEFT) | (d->mb_slice_id[a - 1] == d->mb_slice_id[a] ? FJ_DBK_LEFT : 0));
        if (a >= w) recs[a].dbk = (uint8_t)((recs[a].dbk & ~FJ_DBK_TOP) | (d->mb_slice_id[a - w] == d->mb_slice_id[a] ? FJ_DBK_TOP : 0));
    }
}

static uint8_t *redo_split(HostDec *d, uint8_t **between)
{
    *between = NULL;
    FjHeader *h = (FjHeader *)d->job;
    const uint32_t n = d->pic_size_mbs;
    const size_t used = (size_t)h->coef_off + (size_t)d->coef_blocks * 32u;
    uint8_t *blob = (uint8_t *)malloc(d->job_cap);
    if (!blob) return NULL;
    memcpy(blob, d->job, used);
    memcpy(blob + h->mv_off, d->job + h->mv_off, (size_t)n * 64u);        /* the dense vectors live at the tail of the buffer */
    FjHeader *gh = (FjHeader *)blob;
    FjMbRec *grecs = (FjMbRec *)(blob + gh->rec_off);
    for (uint32_t i = 0; i < d->n_redo; i++) {
        const struct RedoMb *r = &d->redo[i];
        grecs[r->addr] = r->rec;
        memcpy(blob + gh->mv_off + (size_t)r->addr * 64u, r->mv, 64);
    }
    if (d->n_redo2) {
        /* versions written over a first decode and hidden again by later metadata: a job that reconstructs only them */
        uint8_t *mid = (uint8_t *)malloc(d->job_cap);
        if (!mid) { free(blob); return NULL; }
        memcpy(mid, d->job, used);
        memcpy(mid + h->mv_off, d->job + h->mv_off, (size_t)n * 64u);
        FjHeader *mh = (FjHeader *)mid;
        FjMbRec *mrecs = (FjMbRec *)(mid + mh->rec_off);
        for (uint32_t a = 0; a < n; a++) { mrecs[a].pred &= (uint8_t)~FJ_PRED_PHASE2; mrecs[a].dbk = 0; }
        for (uint32_t i = 0; i < d->n_redo2; i++) {
            const struct RedoMb *r = &d->redo2[i];
            mrecs[r->addr] = r->rec;
            mrecs[r->addr].dbk = 0;
            memcpy(mid + mh->mv_off + (size_t)r->addr * 64u, r->mv, 64);
        }
        mh->ghost = 1; mh->dbk_only = 1;
        *between = mid;
    }
    /* concealment comes after everything that was decoded (conceal.c works on the finished picture): all of it moves
     * to the second job, where a synthesised macroblock finds its neighbours as the reference does */
    FjMbRec *recs = (FjMbRec *)(d->job + h->rec_off);
    for (uint32_t a = 0; a < n; a++)
        if (recs[a].kind == FJ_MB_CONCEAL_P || recs[a].kind == FJ_MB_CONCEAL_I) {
            recs[a].pred |= FJ_PRED_PHASE2;
            if (!d->mb_redone || !d->mb_redone[a]) grecs[a].kind = FJ_MB_ABSENT;
        }
    for (uint32_t a = 0; a < n; a++) grecs[a].dbk = 0;          /* deblocking happens once, in the picture's own job */
    gh->ghost = 1;
    h->dbk_only = 1;
    return blob;
}

/* A slice whose data failed to parse: un-decode its macroblocks from the failure point back (an I slice keeps
 * everything up to max(width,10) macroblocks before the last good one, a P slice loses everything) and to its end
 * — reference h264bsdMarkSliceCorrupted, src/h264bsd_slice_data.c:298-354. */
static void mark_slice_corrupted(HostDec *d, uint32_t first_mb)
{
    FjMbRec *recs = (FjMbRec *)(d->job + ((FjHeader *)d->job)->rec_off);
    const uint32_t sid = d->slice_id;
    uint32_t addr = first_mb;
    if (d->last_mb_addr) {
        const uint32_t lim = d->width_mbs > 10 ? d->width_mbs : 10;
        uint32_t i = d->last_mb_addr - 1, cnt = 0;
        while (i > addr) {
            if (d->mb_slice_id[i] == sid && ++cnt >= lim) break;
            i--;
        }
        addr = i;
    }
    if (hd_trace) fprintf(stderr, "TRACE corrupt: first %u last %u start %u sid %u\n", first_mb, d->last_mb_addr, addr, sid);
    d->pic_irregular = 1;
    ghost_store_slice(d, sid);
    /* The coefficient blocks of the macroblocks that become undecoded are reclaimed: they were appended in decoding
     * order behind those of the macroblocks that stay (redundant re-decodes append nothing), so the section is cut at
     * the first of them.  Without this a stream that repeats broken slices over the same macroblocks could grow the
     * section past what job_capacity() reserves (27 blocks per macroblock). */
    uint32_t cut = d->coef_blocks, keep_top = 0;
    do {
        if (hd_trace) fprintf(stderr, "TRACE   mb %u sid %u decoded %u\n", addr, d->mb_slice_id[addr], d->mb_decoded[addr]);
        if (d->mb_slice_id[addr] != sid || !d->mb_decoded[addr]) break;
        /* (a macroblock of this slice that was written over an earlier slice's version — set aside in RedoMb — stays: the
         * picture's macroblock count already includes it, so the picture can complete without concealment and then shows
         * exactly these pixels, filtered with this metadata) */
        if (--d->mb_decoded[addr] == 0 && d->mb_rec_sid[addr] == sid && !(d->mb_redone && d->mb_redone[addr])) {
            if (recs[addr].kind != FJ_MB_ABSENT && recs[addr].coef_idx < cut) cut = recs[addr].coef_idx;
            if (d->mb_ghost && makes_pixels(recs[addr].kind)) { d->mb_ghost[addr] = 1; d->ghost_dirty = 1; }
            recs[addr].kind = FJ_MB_ABSENT;
            d->mb_rec_sid[addr] = 0;
        } else if (d->mb_rec_sid[addr] == sid && recs[addr].kind != FJ_MB_ABSENT) {
            /* a record of this slice that stays: so do its coefficient blocks, wherever they lie among those given back */
            const uint32_t top = recs[addr].coef_idx + rec_blocks(&recs[addr]);
            if (top > keep_top) keep_top = top;
        }
        /* (else, counter at 0 but the record is an earlier slice's: a redundant slice stamped the macroblock and failed
         * before it decoded it — slice_data.c:140 / :322-333.  The macroblock counts as not decoded from here on, but the
         * pixels the earlier slice made stay in the picture unless concealment or a later slice replaces them.) */
        addr = hd_next_mb_in_group(d->slice_group_map, d->pic_size_mbs, addr);
    } while (addr);
    if (keep_top > cut) cut = keep_top < d->coef_blocks ? keep_top : d->coef_blocks;
    d->coef_blocks = cut;
}

/* Plan the concealment of every macroblock that is still not decoded when the access unit ends — reference
 * h264bsdConceal, src/h264bsd_conceal.c:124-260.  The PIXELS are produced on the device: a lost macroblock of a P
 * picture becomes a copy of the co-located macroblock of the first usable reference (FJ_MB_CONCEAL_P, copy list);
 * otherwise it is synthesised from the neighbours decoded or concealed before it in the reference's walking order
 * (FJ_MB_CONCEAL_I, scheduled with the intra macroblocks).  Concealed macroblocks are deblocked as intra, QP 40,
 * offsets 0 (conceal.c:305-313); a wholly lost picture is a copy / grey and is not filtered (:180-206).
 * Returns the number of concealed macroblocks (numErrMbs of the picture). */
static uint32_t plan_concealment(HostDec *d, int p_type)
{
    FjHeader *h = (FjHeader *)d->job;
    FjMbRec *recs = (FjMbRec *)(d->job + h->rec_off);
    int16_t (*mvs)[16][2] = (int16_t (*)[16][2])(d->job + h->mv_off);
    const uint32_t n = d->pic_size_mbs, w = d->width_mbs, hgt = d->height_mbs;
    int ref_slot = -1;
    if (p_type)
        for (uint32_t i = 0; i < 16 && ref_slot < 0; i++) ref_slot = hd_dpb_ref_slot(&d->dpb, i);
    uint32_t first = 0;
    while (first < n && !d->mb_decoded[first]) first++;
    if (hd_trace) { fprintf(stderr, "TRACE conceal p_type %d ref %d decoded:", p_type, ref_slot); for (uint32_t a = 0; a < n; a++) fprintf(stderr, " %u", d->mb_decoded[a]); fprintf(stderr, "\n"); }
    uint32_t seq = 0, count = 0;

#define CONCEAL_ONE(a_, whole_) do { \
        const uint32_t a__ = (a_); \
        FjMbRec *r = &recs[a__]; \
        /* pixels of an earlier slice under a macroblock that a failed redundant slice un-decoded: other macroblocks may \
         * have predicted from them; they are reconstructed first, the concealment replaces them afterwards (RedoMb) */ \
        const int over__ = d->mb_rec_sid[a__] && makes_pixels(r->kind) && !hd_redo_keep_first(d, a__, r, &mvs[a__][0][0]); \
        memset(r, 0, sizeof(*r)); \
        if (over__) r->pred = FJ_PRED_PHASE2; \
        d->mb_rec_sid[a__] = 0; \
        r->qp_y = 40; r->qp_c = 36;          /* QPc of QP 40 with chroma_qp_index_offset 0 */ \
        r->dbk = (whole_) ? 0 : (uint8_t)(FJ_DBK_INNER | ((a__ % w) ? FJ_DBK_LEFT : 0) | (a__ >= w ? FJ_DBK_TOP : 0)); \
        r->coef_idx = seq++; \
        if (ref_slot >= 0) { \
            r->kind = FJ_MB_CONCEAL_P; \
            memset(r->ref_slot, ref_slot, 4); \
        } else { \
            const uint32_t row__ = a__ / w, col__ = a__ % w; \
            r->kind = FJ_MB_CONCEAL_I; \
            r->avail = (uint8_t)((row__ && d->mb_decoded[a__ - w] ? FJ_CONC_ABOVE : 0) | \
                                 (row__ != hgt - 1 && d->mb_decoded[a__ + w] ? FJ_CONC_BELOW : 0) | \
                                 (col__ && d->mb_decoded[a__ - 1] ? FJ_CONC_LEFT : 0) | \
                                 (col__ != w - 1 && d->mb_decoded[a__ + 1] ? FJ_CONC_RIGHT : 0)); \
        } \
        d->mb_decoded[a__] = 1; \
        count++; \
    } while (0)

    if (first == n) {
        /* nothing of the picture survived: no pixel of it is shown or was predicted from by anything that remains, so
         * first and second versions of macroblocks (RedoMb) are of no interest either */
        if (d->n_redo && d->mb_redone) memset(d->mb_redone, 0, n);
        d->n_redo = 0; d->n_redo2 = 0;
        memset(d->mb_rec_sid, 0, (size_t)n * sizeof(uint32_t));
        if (ref_slot >= 0) {
            for (uint32_t a = 0; a < n; a++) CONCEAL_ONE(a, 1);
        } else {
            /* constant 128: every macroblock is an I_PCM that reads the same twelve blocks of 0x80 bytes */
            d->coef_blocks = 0;                      /* no macroblock is left that owns a coefficient block */
            uint8_t *grey = d->job + h->coef_off + (size_t)d->coef_blocks * 32u;
            memset(grey, 128, 384);
            for (uint32_t a = 0; a < n; a++) {
                FjMbRec *r = &recs[a];
                memset(r, 0, sizeof(*r));
                r->kind = FJ_MB_IPCM;
                r->coef_idx = d->coef_blocks;
                d->mb_decoded[a] = 1;
            }
            d->coef_blocks += 12;
            count = n;
        }
        return count;
    }
    const uint32_t row = first / w, col = first % w;
    /* the row of the first good macroblock: leftwards from it, then rightwards */
    for (uint32_t j = col; j--;) CONCEAL_ONE(row * w + j, 0);
    for (uint32_t j = col + 1; j < w; j++) if (!d->mb_decoded[row * w + j]) CONCEAL_ONE(row * w + j, 0);
    /* the rows above it: column by column, upwards */
    if (row)
        for (uint32_t j = 0; j < w; j++)
            for (uint32_t i = row; i--;) CONCEAL_ONE(i * w + j, 0);
    /* the rows below it, in raster order */
    for (uint32_t i = row + 1; i < hgt; i++)
        for (uint32_t j = 0; j < w; j++) if (!d->mb_decoded[i * w + j]) CONCEAL_ONE(i * w + j, 0);
#undef CONCEAL_ONE
    return count;
}

/* ---------------------------------------------------------------- one NAL unit */
int hd_decode(HostDec *d, uint8_t *stream, uint32_t len, uint32_t pic_id, uint32_t *read_bytes)
{
    int rc;
    if (d->prev_buf_not_finished && stream == d->prev_buf_ptr) {
        *read_bytes = d->prev_bytes_consumed;      /* staged payload is still in nal_buf */
    } else {
        rc = hd_extract_nal(d, stream, len, read_bytes);
        if (rc == -2) return HD_MEMALLOC_ERROR;
        if (rc) ERR_RETURN;
        d->prev_bytes_consumed = *read_bytes;
        d->prev_buf_ptr = stream;
    }
    d->prev_buf_not_finished = 0;

    BitReader br = { d->nal_buf, d->nal_size * 8u, 0 };
    if (d->nal_size == 0) ERR_RETURN;
    br_get1(&br);                                                   /* forbidden_zero_bit: not enforced (reference nal_unit.c:77) */
    const int nal_ref_idc = (int)br_get(&br, 2);
    const int nal_type = (int)br_get(&br, 5);
    if (nal_type == 2 || nal_type == 3 || nal_type == 4) ERR_RETURN;   /* data partitioning */
    if ((nal_type == 5 || nal_type == 7 || nal_type == 8) && nal_ref_idc == 0) ERR_RETURN;
    if ((nal_type == 6 || (nal_type >= 9 && nal_type <= 12)) && nal_ref_idc != 0) ERR_RETURN;
    if (nal_type == 0 || nal_type >= 13) return HD_RDY;

    int boundary = 0, conceal_pending = 0;
    rc = check_access_unit_boundary(d, &br, nal_type, nal_ref_idc, &boundary);
    if (rc == -2) return HD_PARAM_SET_ERROR;
    if (rc) ERR_RETURN;

    if (boundary) {
        if (d->pic_started && d->active_sps) {
            /* a picture was left unfinished: conceal what is missing and deliver it (decoder.c:226-266); the NAL
             * unit in hand is decoded on the next call (readBytes 0) */
            if (d->pending_activation) ERR_RETURN;
            *read_bytes = 0;
            d->prev_buf_not_finished = 1;
            d->skip_redundant = 0;
            if (!d->valid_slice_in_au) {
                /* no slice header of the picture was ever decoded: the reference conceals a whole picture — a copy of the
                 * first usable reference, else grey — that is neither stored nor output (decoder.c:242-249, 478:
                 * MarkDecRefPic only for a valid slice).  But it conceals INTO the DPB's spare frame buffer
                 * (h264bsdAllocateDpbImage), and that buffer goes to the next picture or non-existing frame: a later
                 * picture that leaves a macroblock unwritten (FJ_MB_STALE) in that buffer shows these pixels.  So the
                 * pixels are made here too, as a reconstruction-only job into the same buffer. */
                hd_dpb_init_ref_list(&d->dpb);
                if (d->mb) reset_picture_state(d);
                if (d->mb && d->sink.submit && hd_dpb_alloc_current(&d->dpb) >= 0) {
                    if (hd_job_begin(d)) return HD_MEMALLOC_ERROR;
                    (void)plan_concealment(d, 1);
                    d->num_decoded_mbs = d->pic_size_mbs;
                    if (hd_job_finish(d, 0, 1)) ERR_RETURN;
                    ((FjHeader *)d->job)->ghost = 1;
                    if (d->sink.submit(d->sink.user, d->job, ((FjHeader *)d->job)->total_bytes)) ERR_RETURN;
                    tiles_commit(d);
                    reset_picture_state(d);
                }
                (void)hd_decode_poc(&d->poc, d->active_sps, &d->slice, d->cur_nal_type, d->cur_nal_ref_idc);
                d->pic_started = 0;
                return HD_PIC_RDY;
            }
            conceal_pending = 1;
        } else {
            d->valid_slice_in_au = 0;
            d->skip_redundant = 0;
        }
    }

    int pic_ready = 0;
    uint32_t err_mbs = 0;
    if (conceal_pending) {
        err_mbs = plan_concealment(d, d->slice.is_p);
        pic_ready = 1;
    } else
    switch (nal_type) {
    case 7: {
        Sps s;
        if (hd_parse_sps(&br, &s)) ERR_RETURN;
        if (store_sps(d, &s)) return HD_MEMALLOC_ERROR;
        break;
    }
    case 8: {
        Pps p;
        rc = hd_parse_pps(&br, &p);
        if (rc == -2) return HD_MEMALLOC_ERROR;
        if (rc) ERR_RETURN;
        if (store_pps(d, &p)) return HD_MEMALLOC_ERROR;
        break;
    }
    case 1:
    case 5: {
        if (d->skip_redundant) return HD_RDY;
        d->pic_started = 1;
        const int start_of_picture = !d->valid_slice_in_au;
        if (start_of_picture) {
            uint32_t pps_id;
            d->current_pic_id = pic_id;
            if (hd_peek_pps_id(&br, &pps_id)) ERR_RETURN;
            const int old_sps = d->active_sps_id;
            rc = activate_param_sets(d, pps_id, nal_type == 5);
            if (rc) {
                d->active_pps_id = -1; d->active_pps = NULL;
                d->active_sps_id = -1; d->active_sps = NULL;
                d->pending_activation = 0;
                return rc == -2 ? HD_MEMALLOC_ERROR : HD_PARAM_SET_ERROR;
            }
            if (old_sps != d->active_sps_id) {
                /* new sequence: tell the application, expect the same NAL again (decoder.c:343-389) */
                *read_bytes = 0;
                d->prev_buf_not_finished = 1;
                d->old_sps_id = d->active_sps_id;
                return HD_HDRS_RDY;
            }
        }
        if (d->pending_activation) ERR_RETURN;
        SliceHdr sh;
        if (hd_parse_slice_header(&br, &sh, d->active_sps, d->active_pps, nal_type, nal_ref_idc)) ERR_RETURN;
        if (start_of_picture) {
            if (nal_type != 5) {
                if (hd_dpb_check_gaps(&d->dpb, sh.frame_num, nal_ref_idc != 0, d->active_sps->gaps_in_frame_num_allowed))
                    ERR_RETURN;
            }
            if (hd_dpb_alloc_current(&d->dpb) < 0) ERR_RETURN;
            if (hd_job_begin(d)) return HD_MEMALLOC_ERROR;
        }
        if (hd_trace) fprintf(stderr, "TRACE slice nal %d picid %u first_mb %u is_p %d frame_num %u start_of_picture %d redundant %u\n", nal_type, pic_id, sh.first_mb, sh.is_p, sh.frame_num, start_of_picture, sh.redundant_pic_cnt);
        d->slice = sh;
        d->valid_slice_in_au = 1;
        d->cur_nal_type = (uint8_t)nal_type;
        d->cur_nal_ref_idc = (uint8_t)nal_ref_idc;
        hd_slice_group_map(d->slice_group_map, d->active_pps, sh.slice_group_change_cycle, d->width_mbs, d->height_mbs);
        hd_dpb_init_ref_list(&d->dpb);
        if (hd_dpb_reorder_ref_list(&d->dpb, &d->slice)) ERR_RETURN;
        if (hd_decode_slice_data(d, &br, &d->slice, nal_ref_idc)) {
            mark_slice_corrupted(d, d->slice.first_mb);
            ERR_RETURN;
        }
        if (end_of_picture(d)) { pic_ready = 1; d->skip_redundant = 1; }
        break;
    }
    default: break;             /* SEI, AUD, end of sequence/stream, filler: ignored */
    }

    if (!pic_ready) return HD_RDY;

    /* picture complete: queue the frame job, then do the bookkeeping the reference does after
     * deblocking (decoder.c:473-510) */
    const int is_idr = d->cur_nal_type == 5;
    uint8_t *earlier[2] = { NULL, NULL };      /* first decodes; versions written over them (redo_split) */
    fill_undecoded(d);
    if (d->slice_ids_rewritten) restamp_slice_edges(d);
    if (d->n_redo && !(earlier[0] = redo_split(d, &earlier[1]))) ERR_RETURN;
    if (hd_job_finish(d, is_idr, !d->ghost_needed && !earlier[0] && !earlier[1])) { free(earlier[0]); free(earlier[1]); ERR_RETURN; }
    if (d->ghost_needed && ghost_submit(d)) { free(earlier[0]); free(earlier[1]); ERR_RETURN; }
    for (int k = 0, rc = 0; k < 2; k++) {
        if (earlier[k] && !rc) {
            FjHeader *gh = (FjHeader *)earlier[k];
            const FjHeader *mh = (const FjHeader *)d->job;
            rc = fj_finalize(earlier[k], d->job_cap, mh->n_coef_blocks);
            gh->cur_slot = mh->cur_slot; gh->n_slots = mh->n_slots; gh->is_idr = mh->is_idr; gh->pic_seq = mh->pic_seq;
            if (!rc && d->sink.submit && d->sink.submit(d->sink.user, earlier[k], gh->total_bytes)) rc = -1;
        }
        free(earlier[k]);
        if (k == 1 && rc) ERR_RETURN;
    }
    if (d->sink.submit && d->sink.submit(d->sink.user, d->job, ((FjHeader *)d->job)->total_bytes)) {
        fprintf(stderr, "h264bsd-mi355x: frame job submission failed\n");
        ERR_RETURN;
    }
    tiles_commit(d);
    reset_picture_state(d);
    int32_t poc = hd_decode_poc(&d->poc, d->active_sps, &d->slice, d->cur_nal_type, d->cur_nal_ref_idc);
    hd_dpb_mark_current(&d->dpb, &d->slice, d->cur_nal_ref_idc != 0, is_idr, poc, d->current_pic_id, err_mbs);
    d->pic_started = 0;
    d->valid_slice_in_au = 0;
    return HD_PIC_RDY;
}
