/*
 * hd_dpb.c — decoded-picture-buffer bookkeeping: reference marking (H.264 8.2.5), initial
 * reference list + reordering (8.2.4), frame_num gaps (8.2.5.2) and output ordering (C.4).
 *
 * Pure control plane: pictures are identified by SLOT NUMBER; the pixels of slot k live in HBM
 * (engine) and are mirrored to host memory only when the application asks for them.  The slot
 * number is also the reference-picture identity the deblocking kernel compares (the reference
 * compares frame-buffer addresses: src/h264bsd_deblocking.c:349,403).
 *
 * Observable behaviour mirrors the reference (src/h264bsd_dpb.c):
 *  - dpb_size+1 slots, dpb_size = max(num_ref_frames,1) when output reordering is off  (:1014-1040)
 *  - without reordering every picture is queued for output immediately              (:806-815)
 *  - with reordering the smallest-POC picture is bumped while fullness > dpb_size   (:819-823)
 *  - the output queue is emptied at the start of every picture                      (:1260-1261, :681)
 */
#include <string.h>
#include "hostdec.h"

#define NO_LONG_TERM 0xFFFFu

static int is_ref(const DpbPic *p) { return p->status != DPB_UNUSED; }
static int is_short(const DpbPic *p) { return p->status == DPB_SHORT || p->status == DPB_NON_EXISTING; }
static int is_long(const DpbPic *p) { return p->status == DPB_LONG; }

static uint32_t fullness(const Dpb *d)
{
    uint32_t n = 0;
    for (uint32_t i = 0; i < d->n_slots; i++) n += (is_ref(&d->pic[i]) || d->pic[i].to_be_displayed);
    return n;
}
static uint32_t count_refs(const Dpb *d)
{
    uint32_t n = 0;
    for (uint32_t i = 0; i < d->n_slots; i++) n += is_ref(&d->pic[i]);
    return n;
}

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
    memset(d->list, -1, sizeof(d->list));
    return 0;
}

/* next picture in output order -> out queue */
static int bump_one(Dpb *d)
{
    if (d->no_reordering) return -1;
    int best = -1;
    for (uint32_t i = 0; i < d->n_slots; i++)
        if (d->pic[i].to_be_displayed && (best < 0 || d->pic[i].poc < d->pic[best].poc)) best = (int)i;
    if (best < 0) return -1;
    OutPic *o = &d->out[d->n_out++];
    o->slot = (uint8_t)best;
    o->is_idr = d->pic[best].is_idr;
    o->pic_id = d->pic[best].pic_id;
    o->num_err_mbs = d->pic[best].num_err_mbs;
    d->pic[best].to_be_displayed = 0;
    return 0;
}

int hd_dpb_alloc_current(Dpb *d)
{
    for (uint32_t i = 0; i < d->n_slots; i++) {
        if (!is_ref(&d->pic[i]) && !d->pic[i].to_be_displayed) {
            /* do not hand out a slot that still sits un-fetched in the output queue */
            int queued = 0;
            for (uint32_t k = d->out_idx; k < d->n_out; k++) queued |= d->out[k].slot == i;
            if (queued) continue;
            d->cur = (int32_t)i;
            return (int)i;
        }
    }
    d->cur = -1;
    return -1;
}

/* PicNum / LongTermPicNum of every reference relative to the picture being decoded, 8.2.4.1 */
static void set_pic_nums(Dpb *d, uint32_t cur_frame_num)
{
    for (uint32_t i = 0; i < d->n_slots; i++) {
        DpbPic *p = &d->pic[i];
        if (is_short(p))
            p->pic_num = p->frame_num > cur_frame_num ? (int32_t)p->frame_num - (int32_t)d->max_frame_num
                                                      : (int32_t)p->frame_num;
    }
}

/* 8.2.4.2.1: short-term by descending PicNum, then long-term by ascending LongTermPicNum */
void hd_dpb_init_ref_list(Dpb *d)
{
    int n = 0;
    memset(d->list, -1, sizeof(d->list));
    for (uint32_t i = 0; i < d->n_slots; i++)
        if ((int32_t)i != d->cur && is_short(&d->pic[i])) d->list[n++] = (int8_t)i;
    for (int a = 1; a < n; a++) {           /* insertion sort, descending pic_num */
        int8_t s = d->list[a];
        int b = a;
        while (b > 0 && d->pic[d->list[b - 1]].pic_num < d->pic[s].pic_num) { d->list[b] = d->list[b - 1]; b--; }
        d->list[b] = s;
    }
    int first_long = n;
    for (uint32_t i = 0; i < d->n_slots; i++)
        if ((int32_t)i != d->cur && is_long(&d->pic[i])) d->list[n++] = (int8_t)i;
    for (int a = first_long + 1; a < n; a++) {
        int8_t s = d->list[a];
        int b = a;
        while (b > first_long && d->pic[d->list[b - 1]].pic_num > d->pic[s].pic_num) { d->list[b] = d->list[b - 1]; b--; }
        d->list[b] = s;
    }
}

static int find_pic(const Dpb *d, int32_t pic_num, int want_short)
{
    for (uint32_t i = 0; i < d->n_slots; i++) {
        if ((int32_t)i == d->cur) continue;
        const DpbPic *p = &d->pic[i];
        if ((want_short ? is_short(p) : is_long(p)) && p->pic_num == pic_num) return (int)i;
    }
    return -1;
}

/* 8.2.4.3 */
int hd_dpb_reorder_ref_list(Dpb *d, const SliceHdr *sh)
{
    set_pic_nums(d, sh->frame_num);
    hd_dpb_init_ref_list(d);
    if (!sh->reorder_flag) return 0;
    const uint32_t n_active = sh->num_ref_idx_active;
    uint32_t ref_idx = 0;
    int32_t pred = (int32_t)sh->frame_num;
    for (uint32_t c = 0; c < sh->n_reorder; c++) {
        int slot;
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
            slot = find_pic(d, pic_num, 1);
        } else {
            slot = find_pic(d, (int32_t)sh->reorder[c].val, 0);
        }
        if (slot < 0 || d->pic[slot].status == DPB_NON_EXISTING) return -1;
        if (ref_idx >= n_active || n_active >= 32) return -1;
        for (uint32_t j = n_active; j > ref_idx; j--) d->list[j] = d->list[j - 1];
        d->list[ref_idx++] = (int8_t)slot;
        uint32_t k = ref_idx;
        for (uint32_t j = ref_idx; j <= n_active; j++)
            if (d->list[j] != slot) d->list[k++] = d->list[j];
    }
    return 0;
}

/* sliding window, 8.2.5.3 */
static int sliding_window(Dpb *d)
{
    if (count_refs(d) < d->max_ref_frames) return 0;
    int oldest = -1;
    for (uint32_t i = 0; i < d->n_slots; i++)
        if ((int32_t)i != d->cur && is_short(&d->pic[i]) && (oldest < 0 || d->pic[i].pic_num < d->pic[oldest].pic_num))
            oldest = (int)i;
    if (oldest < 0) return -1;
    d->pic[oldest].status = DPB_UNUSED;
    return 0;
}

/* called at the start of every non-IDR picture: empties the output queue and, when the SPS allows
 * gaps, inserts "non-existing" frames for the skipped frame_num values (8.2.5.2) */
int hd_dpb_check_gaps(Dpb *d, uint32_t frame_num, int is_ref_pic, int gaps_allowed)
{
    d->n_out = d->out_idx = 0;
    if (!gaps_allowed) return 0;
    if (frame_num != d->prev_ref_frame_num && frame_num != (d->prev_ref_frame_num + 1) % d->max_frame_num) {
        uint32_t fn = (d->prev_ref_frame_num + 1) % d->max_frame_num;
        do {
            set_pic_nums(d, fn);
            d->cur = -1;
            if (sliding_window(d)) return -1;
            while (fullness(d) >= d->dpb_size)
                if (bump_one(d)) break;
            /* A non-existing frame has no pixels, so it may sit on a slot whose picture was just bumped into
             * the output queue (the picture being decoded may not).  The reference gets the same effect by
             * swapping frame-buffer pointers after the loop (dpb.c:1318-1345); without it a full DPB plus a
             * gap would leave no slot for the current picture. */
            int s = -1;
            for (uint32_t i = 0; i < d->n_slots && s < 0; i++) {
                if (is_ref(&d->pic[i]) || d->pic[i].to_be_displayed) continue;
                for (uint32_t k = d->out_idx; k < d->n_out; k++)
                    if (d->out[k].slot == i) { s = (int)i; break; }
            }
            if (s < 0) s = hd_dpb_alloc_current(d);
            if (s < 0) return -1;
            DpbPic *p = &d->pic[s];
            memset(p, 0, sizeof(*p));
            p->status = DPB_NON_EXISTING;
            p->frame_num = fn;
            p->pic_num = (int32_t)fn;
            d->cur = -1;
            fn = (fn + 1) % d->max_frame_num;
        } while (fn != frame_num);
    } else if (is_ref_pic && frame_num == d->prev_ref_frame_num) {
        return -1;
    }
    if (is_ref_pic) d->prev_ref_frame_num = frame_num;
    else if (frame_num != d->prev_ref_frame_num)
        d->prev_ref_frame_num = (frame_num + d->max_frame_num - 1) % d->max_frame_num;
    return 0;
}

static void drop_all_refs_and_bump(Dpb *d)
{
    for (uint32_t i = 0; i < d->n_slots; i++)
        if ((int32_t)i != d->cur) d->pic[i].status = DPB_UNUSED;
    while (bump_one(d) == 0) {}
    d->max_long_term_idx = NO_LONG_TERM;
    d->prev_ref_frame_num = 0;
}

static void free_long_term_idx(Dpb *d, uint32_t idx)
{
    for (uint32_t i = 0; i < d->n_slots; i++)
        if ((int32_t)i != d->cur && is_long(&d->pic[i]) && (uint32_t)d->pic[i].pic_num == idx) {
            d->pic[i].status = DPB_UNUSED;
            break;
        }
}

/* 8.2.5: marking of the just-decoded picture + output decision. is_ref = nal_ref_idc != 0 */
int hd_dpb_mark_current(Dpb *d, const SliceHdr *sh, int is_ref_pic, int is_idr, int32_t poc,
                        uint32_t pic_id, uint32_t err_mbs)
{
    if (d->cur < 0) return -1;
    DpbPic *cur = &d->pic[d->cur];
    int status = 0;
    const uint8_t display = d->no_reordering ? 0 : 1;
    uint32_t frame_num = sh->frame_num;
    d->last_contains_mmco5 = 0;
    cur->status = DPB_UNUSED;
    cur->to_be_displayed = 0;

    if (!is_ref_pic) {
        cur->frame_num = frame_num;
        cur->pic_num = (int32_t)frame_num;
        cur->poc = poc;
        cur->to_be_displayed = display;
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
    } else {
        int marked_long = 0;
        if (sh->adaptive_marking) {
            for (uint32_t i = 0; i < sh->n_mmco && status == 0; i++) {
                const MmcoCmd *c = &sh->mmco[i];
                int s;
                switch (c->op) {
                case 1:
                    s = find_pic(d, (int32_t)frame_num - (int32_t)c->a, 1);
                    if (s < 0) status = -1; else d->pic[s].status = DPB_UNUSED;
                    break;
                case 2:
                    s = find_pic(d, (int32_t)c->a, 0);
                    if (s < 0) status = -1; else d->pic[s].status = DPB_UNUSED;
                    break;
                case 3:
                    if (d->max_long_term_idx == NO_LONG_TERM || c->b > d->max_long_term_idx) { status = -1; break; }
                    free_long_term_idx(d, c->b);
                    s = find_pic(d, (int32_t)frame_num - (int32_t)c->a, 1);
                    if (s < 0 || d->pic[s].status == DPB_NON_EXISTING) { status = -1; break; }
                    d->pic[s].status = DPB_LONG;
                    d->pic[s].pic_num = (int32_t)c->b;
                    break;
                case 4:
                    d->max_long_term_idx = c->a ? c->a - 1 : NO_LONG_TERM;
                    for (uint32_t k = 0; k < d->n_slots; k++)
                        if ((int32_t)k != d->cur && is_long(&d->pic[k]) &&
                            (d->max_long_term_idx == NO_LONG_TERM || (uint32_t)d->pic[k].pic_num > d->max_long_term_idx))
                            d->pic[k].status = DPB_UNUSED;
                    break;
                case 5:
                    drop_all_refs_and_bump(d);
                    d->last_contains_mmco5 = 1;
                    frame_num = 0;
                    break;
                case 6:
                    if (d->max_long_term_idx == NO_LONG_TERM || c->b > d->max_long_term_idx) { status = -1; break; }
                    free_long_term_idx(d, c->b);
                    if (count_refs(d) < d->max_ref_frames) {
                        cur->frame_num = frame_num;
                        cur->pic_num = (int32_t)c->b;
                        cur->poc = poc;
                        cur->status = DPB_LONG;
                        cur->to_be_displayed = display;
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
            if (count_refs(d) < d->max_ref_frames) {
                cur->frame_num = frame_num;
                cur->pic_num = (int32_t)frame_num;
                cur->poc = poc;
                cur->status = DPB_SHORT;
                cur->to_be_displayed = display;
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
        o->slot = (uint8_t)d->cur;
        o->is_idr = cur->is_idr;
        o->pic_id = pic_id;
        o->num_err_mbs = err_mbs;
    } else {
        while (fullness(d) > d->dpb_size)
            if (bump_one(d)) break;
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
