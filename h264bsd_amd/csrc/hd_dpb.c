/*
 * hd_dpb.c — decoded-picture-buffer bookkeeping: reference marking (H.264 8.2.5), initial
 * reference list + reordering (8.2.4), frame_num gaps (8.2.5.2) and output ordering (C.4).
 *
 * Pure control plane: the pixels of frame-buffer number `slot` live in HBM (engine) and are mirrored to
 * host memory only when the application asks for them.  The slot number is also the reference-picture
 * identity the deblocking kernel compares (the reference compares frame-buffer addresses:
 * src/h264bsd_deblocking.c:349,403).
 *
 * The bookkeeping mirrors the reference's array semantics exactly (src/h264bsd_dpb.c), because several of its
 * observable behaviours depend on WHERE in its `buffer[]` array a picture sits, not only on what it is:
 *   - pic[] is indexed by POSITION like dpbStorage_t.buffer[]; a picture carries its frame buffer (`slot`) with it
 *     when the array is sorted (ShellSort, same increments and comparison as dpb.c:1559-1584 / :139-196, so that
 *     ties end up in the same order);
 *   - the picture being decoded always uses position dpb_size (h264bsdAllocateDpbImage, :905-917);
 *   - RefPicList0 holds positions and is NOT cleared between pictures (h264bsdInitRefPicList only rewrites the
 *     first numRefFrames entries, :1110-1121): a ref_idx beyond the current number of references resolves to
 *     whatever picture sits at the stale position today;
 *   - pictures with equal POC are output in position order (FindSmallestPicOrderCnt, :1381-1409);
 *   - searches look at the first maxRefFrames / numRefFrames positions only; fullness and numRefFrames are running
 *     counters, not recounts.
 * Conforming streams never notice any of this; damaged ones do (tests/test_damaged_streams.py).
 *
 * Observable behaviour otherwise:
 *  - dpb_size+1 positions, dpb_size = max(num_ref_frames,1) when output reordering is off  (:1014-1040)
 *  - without reordering every picture is queued for output immediately              (:806-815)
 *  - with reordering the smallest-POC picture is bumped while fullness > dpb_size   (:819-823)
 *  - the output queue is emptied at the start of every picture                      (:1260-1261, :681)
 */
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include "hostdec.h"

#define NO_LONG_TERM 0xFFFFu

static int is_ref(const DpbPic *p) { return p->status != DPB_UNUSED; }
static int is_short(const DpbPic *p) { return p->status == DPB_SHORT || p->status == DPB_NON_EXISTING; }
static int is_long(const DpbPic *p) { return p->status == DPB_LONG; }
static int is_existing(const DpbPic *p) { return p->status > DPB_NON_EXISTING; }

int hd_dpb_reset(Dpb *d, uint32_t dpb_size, uint32_t max_ref_frames, uint32_t max_frame_num, int no_reordering)
{
    memset(d, 0, sizeof(*d));
    d->max_ref_frames = max_ref_frames ? max_ref_frames : 1;
    d->dpb_size = no_reordering ? d->max_ref_frames : dpb_size;
    if (d->dpb_size < d->max_ref_frames) d->dpb_size = d->max_ref_frames;
    if (d->dpb_size + 1 > FJ_MAX_SLOTS) return -1;
    d->n_slots = d->dpb_size + 1;
    d->max_frame_num = max_frame_num;
    d->no_reordering = (uint8_t)(no_reordering != 0);
    d->max_long_term_idx = NO_LONG_TERM;
    d->cur = -1;
    for (uint32_t i = 0; i < FJ_MAX_SLOTS; i++) d->pic[i].slot = (uint8_t)i;
    memset(d->list, -1, sizeof(d->list));
    return 0;
}

/* ---- the reference's sort of buffer[]: short-term by descending PicNum, long-term by ascending LongTermPicNum,
 *      then pictures only waiting for output, then free entries (ComparePictures, dpb.c:139-196) ---- */
static int compare_pics(const DpbPic *a, const DpbPic *b)
{
    if (!is_ref(a) && !is_ref(b)) {
        if (a->to_be_displayed && !b->to_be_displayed) return -1;
        if (!a->to_be_displayed && b->to_be_displayed) return 1;
        return 0;
    }
    if (!is_ref(b)) return -1;
    if (!is_ref(a)) return 1;
    if (is_short(a) && is_short(b)) return a->pic_num > b->pic_num ? -1 : a->pic_num < b->pic_num ? 1 : 0;
    if (is_short(a)) return -1;
    if (is_short(b)) return 1;
    return a->pic_num > b->pic_num ? 1 : a->pic_num < b->pic_num ? -1 : 0;
}

static void sort_positions(Dpb *d)
{
    const uint32_t n = d->dpb_size + 1;
    for (uint32_t step = 7; step; step >>= 1)
        for (uint32_t i = step; i < n; i++) {
            const DpbPic t = d->pic[i];
            uint32_t j = i;
            while (j >= step && compare_pics(&d->pic[j - step], &t) > 0) {
                d->pic[j] = d->pic[j - step];
                j -= step;
            }
            d->pic[j] = t;
        }
}

/* next picture in output order -> out queue (OutputPicture, dpb.c:1424-1458) */
static int bump_one(Dpb *d)
{
    if (d->no_reordering) return -1;
    int best = -1;
    for (uint32_t i = 0; i <= d->dpb_size; i++)
        if (d->pic[i].to_be_displayed && (best < 0 || d->pic[i].poc < d->pic[best].poc)) best = (int)i;
    if (best < 0) return -1;
    OutPic *o = &d->out[d->n_out++];
    o->slot = d->pic[best].slot;
    o->is_idr = d->pic[best].is_idr;
    o->pic_id = d->pic[best].pic_id;
    o->num_err_mbs = d->pic[best].num_err_mbs;
    d->pic[best].to_be_displayed = 0;
    if (!is_ref(&d->pic[best])) d->fullness--;
    return 0;
}

/* h264bsdAllocateDpbImage: the picture being decoded always takes the last position */
int hd_dpb_alloc_current(Dpb *d)
{
    d->cur = (int32_t)d->dpb_size;
    return d->pic[d->cur].slot;
}

int hd_dpb_cur_slot(const Dpb *d) { return d->cur < 0 ? -1 : d->pic[d->cur].slot; }

/* h264bsdGetRefPicData: the frame buffer behind RefPicList0[ref_idx], -1 when there is none / it does not exist */
int hd_dpb_ref_slot(const Dpb *d, uint32_t ref_idx)
{
    if (ref_idx > 16 || d->list[ref_idx] < 0) return -1;
    const DpbPic *p = &d->pic[d->list[ref_idx]];
    return is_existing(p) ? p->slot : -1;
}

/* PicNum / LongTermPicNum of every reference relative to the picture being decoded, 8.2.4.1 (SetPicNums) */
static void set_pic_nums(Dpb *d, uint32_t cur_frame_num)
{
    for (uint32_t i = 0; i < d->num_ref_frames; i++) {
        DpbPic *p = &d->pic[i];
        if (is_short(p))
            p->pic_num = p->frame_num > cur_frame_num ? (int32_t)p->frame_num - (int32_t)d->max_frame_num
                                                      : (int32_t)p->frame_num;
    }
}

/* h264bsdInitRefPicList: the sorted array IS the initial list; entries beyond numRefFrames keep their old value */
void hd_dpb_init_ref_list(Dpb *d)
{
    for (uint32_t i = 0; i < d->num_ref_frames; i++) d->list[i] = (int8_t)i;
}

/* FindDpbPic */
static int find_pic(const Dpb *d, int32_t pic_num, int want_short)
{
    for (uint32_t i = 0; i < d->max_ref_frames; i++) {
        const DpbPic *p = &d->pic[i];
        if ((want_short ? is_short(p) : is_long(p)) && p->pic_num == pic_num) return (int)i;
    }
    return -1;
}

int hd_dpb_reorder_ref_list(Dpb *d, const SliceHdr *sh)
{
    set_pic_nums(d, sh->frame_num);
    if (!sh->reorder_flag) return 0;
    const uint32_t n_active = sh->num_ref_idx_active;
    uint32_t ref_idx = 0;
    int32_t pred = (int32_t)sh->frame_num;
    for (uint32_t c = 0; c < sh->n_reorder; c++) {
        int pos;
        if (sh->reorder[c].idc < 2) {
            int32_t no_wrap;
            if (sh->reorder[c].idc == 0) {
                no_wrap = pred - (int32_t)sh->reorder[c].val;
                if (no_wrap < 0) no_wrap += (int32_t)d->max_frame_num;
            } else {
                no_wrap = pred + (int32_t)sh->reorder[c].val;
                if (no_wrap >= (int32_t)d->max_frame_num) no_wrap -= (int32_t)d->max_frame_num;
            }
            pred = no_wrap;
            int32_t pic_num = no_wrap > (int32_t)sh->frame_num ? no_wrap - (int32_t)d->max_frame_num : no_wrap;
            pos = find_pic(d, pic_num, 1);
        } else {
            pos = find_pic(d, (int32_t)sh->reorder[c].val, 0);
        }
        if (pos < 0 || !is_existing(&d->pic[pos])) return -1;
        if (ref_idx >= n_active || n_active >= 32) return -1;
        for (uint32_t j = n_active; j > ref_idx; j--) d->list[j] = d->list[j - 1];
        d->list[ref_idx++] = (int8_t)pos;
        uint32_t k = ref_idx;
        for (uint32_t j = ref_idx; j <= n_active; j++)
            if (d->list[j] != pos) d->list[k++] = d->list[j];
    }
    return 0;
}

static void set_unused(Dpb *d, uint32_t pos)
{
    d->pic[pos].status = DPB_UNUSED;
    d->num_ref_frames--;
    if (!d->pic[pos].to_be_displayed) d->fullness--;
}

/* sliding window, 8.2.5.3 (SlidingWindowRefPicMarking) */
static int sliding_window(Dpb *d)
{
    if (d->num_ref_frames < d->max_ref_frames) return 0;
    int oldest = -1;
    int32_t pic_num = 0;
    for (uint32_t i = 0; i < d->num_ref_frames; i++)
        if (is_short(&d->pic[i]) && (d->pic[i].pic_num < pic_num || oldest < 0)) {
            oldest = (int)i;
            pic_num = d->pic[i].pic_num;
        }
    if (oldest < 0) return -1;
    set_unused(d, (uint32_t)oldest);
    return 0;
}

/* called at the start of every non-IDR picture: empties the output queue and, when the SPS allows
 * gaps, inserts "non-existing" frames for the skipped frame_num values (8.2.5.2; h264bsdCheckGapsInFrameNum) */
int hd_dpb_check_gaps(Dpb *d, uint32_t frame_num, int is_ref_pic, int gaps_allowed)
{
    d->n_out = d->out_idx = 0;
    if (!gaps_allowed) return 0;
    if (frame_num != d->prev_ref_frame_num && frame_num != (d->prev_ref_frame_num + 1) % d->max_frame_num) {
        uint32_t fn = (d->prev_ref_frame_num + 1) % d->max_frame_num;
        /* frame buffer of the free last position: the picture being decoded must not end up on a buffer that the
         * loop below hands to the output queue (dpb.c:1283-1345) */
        const uint8_t spare = d->pic[d->dpb_size].slot;
        do {
            set_pic_nums(d, fn);
            if (sliding_window(d)) return -1;
            while (d->fullness >= d->dpb_size)
                if (bump_one(d)) break;
            DpbPic *p = &d->pic[d->dpb_size];
            p->status = DPB_NON_EXISTING;
            p->frame_num = fn;
            p->pic_num = (int32_t)fn;
            p->poc = 0;
            p->to_be_displayed = 0;
            d->fullness++;
            d->num_ref_frames++;
            sort_positions(d);
            fn = (fn + 1) % d->max_frame_num;
        } while (fn != frame_num);
        if (d->n_out) {
            for (uint32_t i = 0; i < d->n_out; i++)
                if (d->out[i].slot == d->pic[d->dpb_size].slot) {
                    for (uint32_t k = 0; k < d->dpb_size; k++)
                        if (d->pic[k].slot == spare) {
                            d->pic[k].slot = d->pic[d->dpb_size].slot;
                            d->pic[d->dpb_size].slot = spare;
                            break;
                        }
                    break;
                }
        }
    } else if (is_ref_pic && frame_num == d->prev_ref_frame_num) {
        return -1;
    }
    if (is_ref_pic) d->prev_ref_frame_num = frame_num;
    else if (frame_num != d->prev_ref_frame_num)
        d->prev_ref_frame_num = (frame_num + d->max_frame_num - 1) % d->max_frame_num;
    return 0;
}

/* Mmcop5 */
static void drop_all_refs_and_bump(Dpb *d)
{
    for (uint32_t i = 0; i < 16; i++)
        if (is_ref(&d->pic[i])) {
            d->pic[i].status = DPB_UNUSED;
            if (!d->pic[i].to_be_displayed) d->fullness--;
        }
    while (bump_one(d) == 0) {}
    d->num_ref_frames = 0;
    d->max_long_term_idx = NO_LONG_TERM;
    d->prev_ref_frame_num = 0;
}

static void free_long_term_idx(Dpb *d, uint32_t idx)
{
    for (uint32_t i = 0; i < d->max_ref_frames; i++)
        if (is_long(&d->pic[i]) && (uint32_t)d->pic[i].pic_num == idx) {
            set_unused(d, i);
            break;
        }
}

/* 8.2.5: marking of the just-decoded picture + output decision (h264bsdMarkDecRefPic). is_ref = nal_ref_idc != 0 */
int hd_dpb_mark_current(Dpb *d, const SliceHdr *sh, int is_ref_pic, int is_idr, int32_t poc,
                        uint32_t pic_id, uint32_t err_mbs)
{
    if (d->cur < 0) return -1;
    DpbPic *cur = &d->pic[d->cur];
    int status = 0;
    const uint8_t display = d->no_reordering ? 0 : 1;
    uint32_t frame_num = sh->frame_num;
    d->last_contains_mmco5 = 0;

    if (!is_ref_pic) {
        cur->status = DPB_UNUSED;
        cur->frame_num = frame_num;
        cur->pic_num = (int32_t)frame_num;
        cur->poc = poc;
        cur->to_be_displayed = display;
        if (!d->no_reordering) d->fullness++;
    } else if (is_idr) {
        d->n_out = d->out_idx = 0;
        drop_all_refs_and_bump(d);
        if (sh->no_output_of_prior_pics || d->no_reordering) d->n_out = d->out_idx = 0;
        cur->status = sh->long_term_reference_flag ? DPB_LONG : DPB_SHORT;
        d->max_long_term_idx = sh->long_term_reference_flag ? 0 : NO_LONG_TERM;
        cur->frame_num = 0;
        cur->pic_num = 0;
        cur->poc = 0;
        cur->to_be_displayed = display;
        d->fullness = 1;
        d->num_ref_frames = 1;
    } else {
        int marked_long = 0;
        if (sh->adaptive_marking) {
            for (uint32_t i = 0; i < sh->n_mmco && status == 0; i++) {
                const MmcoCmd *c = &sh->mmco[i];
                int s;
                switch (c->op) {
                case 1:
                    s = find_pic(d, (int32_t)frame_num - (int32_t)c->a, 1);
                    if (s < 0) status = -1; else set_unused(d, (uint32_t)s);
                    break;
                case 2:
                    s = find_pic(d, (int32_t)c->a, 0);
                    if (s < 0) status = -1; else set_unused(d, (uint32_t)s);
                    break;
                case 3:
                    if (d->max_long_term_idx == NO_LONG_TERM || c->b > d->max_long_term_idx) { status = -1; break; }
                    free_long_term_idx(d, c->b);
                    s = find_pic(d, (int32_t)frame_num - (int32_t)c->a, 1);
                    if (s < 0 || !is_existing(&d->pic[s])) { status = -1; break; }
                    d->pic[s].status = DPB_LONG;
                    d->pic[s].pic_num = (int32_t)c->b;
                    break;
                case 4:
                    d->max_long_term_idx = c->a ? c->a - 1 : NO_LONG_TERM;
                    for (uint32_t k = 0; k < d->max_ref_frames; k++)
                        if (is_long(&d->pic[k]) &&
                            (d->max_long_term_idx == NO_LONG_TERM || (uint32_t)d->pic[k].pic_num > d->max_long_term_idx))
                            set_unused(d, k);
                    break;
                case 5:
                    drop_all_refs_and_bump(d);
                    d->last_contains_mmco5 = 1;
                    frame_num = 0;
                    break;
                case 6:
                    if (d->max_long_term_idx == NO_LONG_TERM || c->b > d->max_long_term_idx) { status = -1; break; }
                    free_long_term_idx(d, c->b);
                    if (d->num_ref_frames < d->max_ref_frames) {
                        cur->frame_num = frame_num;
                        cur->pic_num = (int32_t)c->b;
                        cur->poc = poc;
                        cur->status = DPB_LONG;
                        cur->to_be_displayed = display;
                        d->num_ref_frames++;
                        d->fullness++;
                        marked_long = 1;
                    } else status = -1;
                    break;
                default: status = -1; break;
                }
            }
        } else {
            status = sliding_window(d);
        }
        if (!marked_long) {
            if (d->num_ref_frames < d->max_ref_frames) {
                cur->frame_num = frame_num;
                cur->pic_num = (int32_t)frame_num;
                cur->poc = poc;
                cur->status = DPB_SHORT;
                cur->to_be_displayed = display;
                d->fullness++;
                d->num_ref_frames++;
            } else {
                status = -1;
            }
        }
    }
    cur->is_idr = (uint8_t)is_idr;
    cur->pic_id = pic_id;
    cur->num_err_mbs = err_mbs;

    if (d->no_reordering) {
        OutPic *o = &d->out[d->n_out++];
        o->slot = cur->slot;
        o->is_idr = cur->is_idr;
        o->pic_id = pic_id;
        o->num_err_mbs = err_mbs;
    } else {
        while (d->fullness > d->dpb_size)
            if (bump_one(d)) break;
    }
    sort_positions(d);
    if (hd_trace) {
        fprintf(stderr, "TRACE dpb size %u full %u numRef %u numOut %u:", d->dpb_size, d->fullness, d->num_ref_frames, d->n_out);
        for (uint32_t i = 0; i <= d->dpb_size; i++)
            if (d->pic[i].status || d->pic[i].to_be_displayed)
                fprintf(stderr, " [%u st%d disp%d fn%u pn%d poc%d id%u]", i, d->pic[i].status, d->pic[i].to_be_displayed,
                        d->pic[i].frame_num, d->pic[i].pic_num, d->pic[i].poc, d->pic[i].pic_id);
        fprintf(stderr, "\n");
    }
    d->cur = -1;
    return status;
}

void hd_dpb_flush(Dpb *d)
{
    if (!d->n_slots) return;
    while (bump_one(d) == 0) {}
}

const OutPic *hd_dpb_next_output(Dpb *d)
{
    if (d->out_idx < d->n_out) return &d->out[d->out_idx++];
    return NULL;
}
