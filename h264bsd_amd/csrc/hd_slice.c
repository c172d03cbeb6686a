/*
 * hd_slice.c — slice header (H.264 7.3.3 / 7.4.3), picture order count (8.2.1) and the
 * macroblock-to-slice-group map (8.2.2).
 *
 * Baseline subset, frames only.  Accept/reject rules follow the reference
 * (src/h264bsd_slice_header.c:120-380: only P (0/5) and I (2/7) slices, frame_num must be 0 on IDR,
 * filter offsets stored doubled :345,:355; POC: src/h264bsd_pic_order_cnt.c:78-347 returns
 * min(top, bottom) and zeroes it for pictures carrying MMCO5).
 */
#include <string.h>
#include "hostdec.h"

static uint32_t bits_for(uint32_t n) /* ceil(log2(n)) for n >= 1 */
{
    uint32_t b = 0;
    while ((1u << b) < n) b++;
    return b;
}
static uint32_t log2_exact(uint32_t pow2)
{
    uint32_t b = 0;
    while ((1u << b) < pow2) b++;
    return b;
}

int hd_peek_pps_id(const BitReader *br0, uint32_t *pps_id)
{
    BitReader br = *br0;
    br_ue(&br);                 /* first_mb_in_slice */
    br_ue(&br);                 /* slice_type */
    uint32_t v = br_ue(&br);
    if (br_overrun(&br) || v >= HD_MAX_PPS) return -1;
    *pps_id = v;
    return 0;
}

int hd_parse_slice_header(BitReader *br, SliceHdr *sh, const Sps *sps, const Pps *pps,
                          int nal_type, int nal_ref_idc)
{
    memset(sh, 0, sizeof(*sh));
    const int is_idr = nal_type == 5;
    const uint32_t pic_mbs = sps->width_mbs * sps->height_mbs;

    sh->first_mb = br_ue(br);
    if (sh->first_mb >= pic_mbs) return -1;
    uint32_t st = br_ue(br);
    if (st == 0 || st == 5) sh->is_p = 1;
    else if (st == 2 || st == 7) sh->is_p = 0;
    else return -1;
    if (is_idr && sh->is_p) return -1;
    if (sh->is_p && sps->num_ref_frames == 0) return -1;
    sh->pps_id = br_ue(br);
    if (sh->pps_id != pps->pps_id) return -1;
    sh->frame_num = br_get(br, log2_exact(sps->max_frame_num));
    if (is_idr && sh->frame_num != 0) return -1;
    if (is_idr) {
        sh->idr_pic_id = br_ue(br);
        if (sh->idr_pic_id > 65535) return -1;
    }
    if (sps->poc_type == 0) {
        sh->poc_lsb = br_get(br, log2_exact(sps->max_poc_lsb));
        if (pps->pic_order_present) sh->delta_poc_bottom = br_se(br);
        if (is_idr) {   /* an IDR frame must end up with PicOrderCnt 0 */
            int32_t top = (int32_t)sh->poc_lsb, bot = top + sh->delta_poc_bottom;
            if (sh->poc_lsb > sps->max_poc_lsb / 2 || (top < bot ? top : bot) != 0) return -1;
        }
    } else if (sps->poc_type == 1 && !sps->delta_pic_order_always_zero) {
        sh->delta_poc[0] = br_se(br);
        if (pps->pic_order_present) sh->delta_poc[1] = br_se(br);
        if (is_idr) {
            int32_t top = sh->delta_poc[0], bot = top + sps->offset_for_top_to_bottom_field + sh->delta_poc[1];
            if ((top < bot ? top : bot) != 0) return -1;
        }
    }
    if (pps->redundant_pic_cnt_present) {
        sh->redundant_pic_cnt = br_ue(br);
        if (sh->redundant_pic_cnt > 127) return -1;
    }
    sh->num_ref_idx_active = pps->num_ref_idx_l0_active;
    if (sh->is_p) {
        if (br_get1(br)) {
            uint32_t v = br_ue(br);
            if (v > 15) return -1;
            sh->num_ref_idx_active = v + 1;
        } else if (sh->num_ref_idx_active > 16) {
            return -1;
        }
        /* ref_pic_list_reordering() */
        sh->reorder_flag = (uint8_t)br_get1(br);
        if (sh->reorder_flag) {
            for (;;) {
                /* the reference counts the terminating command as well and tests BEFORE it reads the next one
                 * (slice_header.c:496-501): at most num_ref_idx_active commands, the end marker no later than
                 * position num_ref_idx_active */
                if (sh->n_reorder > sh->num_ref_idx_active) return -1;
                uint32_t idc = br_ue(br);
                if (br_overrun(br) || idc > 3) return -1;
                if (idc == 3) break;
                uint32_t v = br_ue(br);
                if (idc < 2) { if (v >= sps->max_frame_num) return -1; v += 1; }
                sh->reorder[sh->n_reorder].idc = (uint8_t)idc;
                sh->reorder[sh->n_reorder].val = v;
                sh->n_reorder++;
            }
            if (sh->n_reorder == 0) return -1;
        }
    }
    if (nal_ref_idc != 0) {
        if (is_idr) {
            sh->no_output_of_prior_pics = (uint8_t)br_get1(br);
            sh->long_term_reference_flag = (uint8_t)br_get1(br);
            if (sh->long_term_reference_flag && sps->num_ref_frames == 0) return -1;
        } else {
            sh->adaptive_marking = (uint8_t)br_get1(br);
            if (sh->adaptive_marking) {
                uint32_t n4 = 0, n5 = 0, n6 = 0, n123 = 0;
                for (uint32_t i = 0;; i++) {
                    /* the count is checked before every operation is read, the terminating 0 included
                     * (slice_header.c:618-624) */
                    if (i > 2u * sps->num_ref_frames + 2u) return -1;
                    uint32_t op = br_ue(br);
                    if (br_overrun(br) || op > 6) return -1;
                    if (op == 0) break;
                    if (sh->n_mmco >= 35) return -1;
                    MmcoCmd *c = &sh->mmco[sh->n_mmco++];
                    c->op = (uint8_t)op;
                    if (op == 1 || op == 3) c->a = br_ue(br) + 1;          /* difference_of_pic_nums */
                    if (op == 2) c->a = br_ue(br);                        /* long_term_pic_num */
                    if (op == 3 || op == 6) c->b = br_ue(br);             /* long_term_frame_idx */
                    if (op == 4) { c->a = br_ue(br); if (c->a > sps->num_ref_frames) return -1; }   /* max_long_term_frame_idx_plus1, in [0, num_ref_frames] (slice_header.c:668-673) */
                    if (op == 4) n4++;
                    if (op == 5) n5++;
                    if (op == 6) n6++;
                    if (op >= 1 && op <= 3) n123++;
                }
                if (n4 > 1 || n5 > 1 || n6 > 1 || (n123 && n5)) return -1;
            }
        }
    }
    sh->slice_qp_delta = br_se(br);
    {
        int32_t qp = pps->pic_init_qp + sh->slice_qp_delta;
        if (qp < 0 || qp > 51) return -1;
    }
    if (pps->deblocking_filter_control_present) {
        uint32_t idc = br_ue(br);
        if (idc > 2) return -1;
        sh->disable_deblocking_filter_idc = (uint8_t)idc;
        if (idc != 1) {
            int32_t a = br_se(br), b = br_se(br);
            if (a < -6 || a > 6 || b < -6 || b > 6) return -1;
            sh->alpha_off = 2 * a;
            sh->beta_off = 2 * b;
        }
    }
    if (pps->num_slice_groups > 1 && pps->slice_group_map_type >= 3 && pps->slice_group_map_type <= 5) {
        uint32_t units = pic_mbs / pps->slice_group_change_rate;
        if (pic_mbs % pps->slice_group_change_rate) units++;
        uint32_t nb = bits_for(units + 1);
        sh->slice_group_change_cycle = br_get(br, nb);
        if (sh->slice_group_change_cycle > units) return -1;
    }
    if (br_overrun(br)) return -1;
    return 0;
}

/* 8.2.1 for frames; returns PicOrderCnt(CurrPic) = min(top, bottom) */
int32_t hd_decode_poc(PocState *st, const Sps *sps, const SliceHdr *sh, int nal_type, int nal_ref_idc)
{
    const int is_idr = nal_type == 5;
    int mmco5 = 0;
    if (sh->adaptive_marking)
        for (uint32_t i = 0; i < sh->n_mmco; i++) if (sh->mmco[i].op == 5) mmco5 = 1;

    int32_t poc;
    if (sps->poc_type == 0) {
        if (is_idr) { st->prev_poc_msb = 0; st->prev_poc_lsb = 0; }
        int32_t msb = st->prev_poc_msb;
        if (sh->poc_lsb < st->prev_poc_lsb && st->prev_poc_lsb - sh->poc_lsb >= sps->max_poc_lsb / 2)
            msb += (int32_t)sps->max_poc_lsb;
        else if (sh->poc_lsb > st->prev_poc_lsb && sh->poc_lsb - st->prev_poc_lsb > sps->max_poc_lsb / 2)
            msb -= (int32_t)sps->max_poc_lsb;
        int32_t top = msb + (int32_t)sh->poc_lsb;
        poc = sh->delta_poc_bottom < 0 ? top + sh->delta_poc_bottom : top;
        if (nal_ref_idc) {
            if (mmco5) {
                /* after MMCO5 the stored top-field count is top - min(top,bottom) */
                st->prev_poc_msb = 0;
                st->prev_poc_lsb = sh->delta_poc_bottom < 0 ? (uint32_t)(-sh->delta_poc_bottom) : 0;
                poc = 0;
            } else {
                st->prev_poc_msb = msb;
                st->prev_poc_lsb = sh->poc_lsb;
            }
        }
        return poc;
    }

    uint32_t frame_num_offset;
    if (is_idr) frame_num_offset = 0;
    else if (st->prev_frame_num > sh->frame_num) frame_num_offset = st->prev_frame_num_offset + sps->max_frame_num;
    else frame_num_offset = st->prev_frame_num_offset;

    if (sps->poc_type == 1) {
        uint32_t abs_frame_num = sps->num_ref_frames_in_poc_cycle ? frame_num_offset + sh->frame_num : 0;
        if (nal_ref_idc == 0 && abs_frame_num > 0) abs_frame_num--;
        int32_t expected = 0;
        if (abs_frame_num > 0) {
            uint32_t cycles = (abs_frame_num - 1) / sps->num_ref_frames_in_poc_cycle;
            uint32_t in_cycle = (abs_frame_num - 1) % sps->num_ref_frames_in_poc_cycle;
            int32_t per_cycle = 0;
            for (uint32_t i = 0; i < sps->num_ref_frames_in_poc_cycle; i++) per_cycle += sps->offset_for_ref_frame[i];
            expected = (int32_t)cycles * per_cycle;
            for (uint32_t i = 0; i <= in_cycle; i++) expected += sps->offset_for_ref_frame[i];
        }
        if (nal_ref_idc == 0) expected += sps->offset_for_non_ref_pic;
        poc = expected + sh->delta_poc[0];
        int32_t to_bottom = sps->offset_for_top_to_bottom_field + sh->delta_poc[1];
        if (to_bottom < 0) poc += to_bottom;
    } else {
        if (is_idr) poc = 0;
        else poc = 2 * (int32_t)(frame_num_offset + sh->frame_num) - (nal_ref_idc == 0 ? 1 : 0);
    }
    if (mmco5) {
        st->prev_frame_num_offset = 0;
        st->prev_frame_num = 0;
        poc = 0;
    } else {
        st->prev_frame_num_offset = frame_num_offset;
        st->prev_frame_num = sh->frame_num;
    }
    return poc;
}

/* 8.2.2: mapUnitToSliceGroupMap for frame_mbs_only pictures (map unit == macroblock) */
void hd_slice_group_map(uint32_t *map, const Pps *pps, uint32_t change_cycle, uint32_t w, uint32_t h)
{
    const uint32_t n = w * h, ng = pps->num_slice_groups;
    if (ng <= 1) { memset(map, 0, n * sizeof(uint32_t)); return; }

    uint32_t units0 = 0, size_upper_left = 0;
    if (pps->slice_group_map_type >= 3 && pps->slice_group_map_type <= 5) {
        units0 = change_cycle * pps->slice_group_change_rate;
        if (units0 > n) units0 = n;
        size_upper_left = pps->slice_group_change_direction ? n - units0 : units0;
    }
    switch (pps->slice_group_map_type) {
    case 0: { /* interleaved */
        uint32_t i = 0;
        while (i < n)
            for (uint32_t g = 0; g < ng && i < n; i += pps->run_length[g++])
                for (uint32_t j = 0; j < pps->run_length[g] && i + j < n; j++) map[i + j] = g;
        break;
    }
    case 1: /* dispersed */
        for (uint32_t i = 0; i < n; i++) map[i] = ((i % w) + (((i / w) * ng) / 2)) % ng;
        break;
    case 2: /* foreground + leftover */
        for (uint32_t i = 0; i < n; i++) map[i] = ng - 1;
        for (int32_t g = (int32_t)ng - 2; g >= 0; g--) {
            uint32_t tl = pps->top_left[g], brr = pps->bottom_right[g];
            if (tl > brr || brr >= n || (tl % w) > (brr % w)) continue;
            for (uint32_t y = tl / w; y <= brr / w; y++)
                for (uint32_t x = tl % w; x <= brr % w; x++) map[y * w + x] = (uint32_t)g;
        }
        break;
    case 3: { /* box-out */
        const int dir = pps->slice_group_change_direction;
        for (uint32_t i = 0; i < n; i++) map[i] = 1;
        int32_t x = ((int32_t)w - dir) / 2, y = ((int32_t)h - dir) / 2;
        int32_t left = x, top = y, right = x, bottom = y;
        int32_t xd = dir - 1, yd = dir;
        for (uint32_t k = 0; k < units0;) {
            uint32_t at = (uint32_t)y * w + (uint32_t)x;
            uint32_t fresh = map[at] == 1;
            if (fresh) map[at] = 0;
            if (xd == -1 && x == left) {
                left = left > 0 ? left - 1 : 0; x = left; xd = 0; yd = 2 * dir - 1;
            } else if (xd == 1 && x == right) {
                right = right + 1 < (int32_t)w ? right + 1 : (int32_t)w - 1; x = right; xd = 0; yd = 1 - 2 * dir;
            } else if (yd == -1 && y == top) {
                top = top > 0 ? top - 1 : 0; y = top; xd = 1 - 2 * dir; yd = 0;
            } else if (yd == 1 && y == bottom) {
                bottom = bottom + 1 < (int32_t)h ? bottom + 1 : (int32_t)h - 1; y = bottom; xd = 2 * dir - 1; yd = 0;
            } else { x += xd; y += yd; }
            k += fresh;
        }
        break;
    }
    case 4: /* raster scan */
        for (uint32_t i = 0; i < n; i++)
            map[i] = i < size_upper_left ? pps->slice_group_change_direction : 1u - pps->slice_group_change_direction;
        break;
    case 5: { /* wipe */
        uint32_t k = 0;
        for (uint32_t x = 0; x < w; x++)
            for (uint32_t y = 0; y < h; y++)
                map[y * w + x] = k++ < size_upper_left ? pps->slice_group_change_direction
                                                       : 1u - pps->slice_group_change_direction;
        break;
    }
    default: /* 6: explicit */
        for (uint32_t i = 0; i < n; i++)
            map[i] = (pps->slice_group_id && i < pps->pic_size_in_map_units) ? pps->slice_group_id[i] : 0;
        break;
    }
}
