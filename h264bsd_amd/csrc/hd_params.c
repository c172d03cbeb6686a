/*
 * hd_params.c — sequence / picture parameter sets and the VUI subset the API exposes
 * (H.264 7.3.2.1, 7.3.2.2, E.1.1, Table A-1).
 *
 * Accepts what the reference accepts (src/h264bsd_seq_param_set.c, src/h264bsd_pic_param_set.c,
 * src/h264bsd_vui.c): baseline syntax only, frame_mbs_only_flag must be 1, CAVLC only, no weighted
 * prediction; DPB size from the level table with the num_ref_frames fallback and the VUI
 * max_dec_frame_buffering override (seq_param_set.c:304-358).
 */
#include <stdlib.h>
#include <string.h>
#include "hostdec.h"

#define CHECK(b) do { if (br_overrun(b)) return -1; } while (0)

static uint32_t level_dpb_frames(uint32_t pic_mbs, uint32_t level_idc)
{
    /* Table A-1: MaxDPB (bytes, = 1024 * MaxDPB[kB]) and MaxFS; 0 = unknown level */
    uint32_t bytes, max_fs;
    switch (level_idc) {
    case 10: bytes = 152064;  max_fs = 99; break;
    case 11: bytes = 345600;  max_fs = 396; break;
    case 12: case 13: case 20: bytes = 912384; max_fs = 396; break;
    case 21: bytes = 1824768; max_fs = 792; break;
    case 22: case 30: bytes = 3110400; max_fs = 1620; break;
    case 31: bytes = 6912000; max_fs = 3600; break;
    case 32: bytes = 7864320; max_fs = 5120; break;
    case 40: case 41: bytes = 12582912; max_fs = 8192; break;
    case 42: bytes = 13369344; max_fs = 8704; break;
    case 50: bytes = 42393600; max_fs = 22080; break;
    case 51: bytes = 70778880; max_fs = 36864; break;
    default: return 0xFFFFFFFFu;
    }
    if (pic_mbs > max_fs) return 0xFFFFFFFFu;
    uint32_t n = bytes / (pic_mbs * 384u);
    return n < 16 ? n : 16;
}

static int skip_hrd(BitReader *br)
{
    uint32_t cpb_cnt = br_ue(br) + 1;
    if (cpb_cnt > 32) return -1;
    br_get(br, 4); br_get(br, 4);
    for (uint32_t i = 0; i < cpb_cnt; i++) {
        /* bit_rate_value_minus1, cpb_size_value_minus1: [0, 2^32 - 2] (vui.c:437-456) */
        if (br_ue(br) > 4294967294u || br_ue(br) > 4294967294u) return -1;
        br_get1(br);
    }
    br_get(br, 5); br_get(br, 5); br_get(br, 5); br_get(br, 5);
    CHECK(br);
    return 0;
}

static int parse_vui(BitReader *br, Sps *s)
{
    s->aspect_ratio_present = (uint8_t)br_get1(br);
    if (s->aspect_ratio_present) {
        s->aspect_ratio_idc = (uint8_t)br_get(br, 8);
        if (s->aspect_ratio_idc == 255) { s->sar_width = br_get(br, 16); s->sar_height = br_get(br, 16); }
    }
    if (br_get1(br)) br_get1(br);                       /* overscan */
    s->video_signal_type_present = (uint8_t)br_get1(br);
    s->matrix_coefficients = 2;
    if (s->video_signal_type_present) {
        br_get(br, 3);                                  /* video_format */
        s->video_full_range = (uint8_t)br_get1(br);
        s->colour_description_present = (uint8_t)br_get1(br);
        if (s->colour_description_present) {
            br_get(br, 8); br_get(br, 8);
            s->matrix_coefficients = (uint8_t)br_get(br, 8);
        }
    }
    /* the reference rejects the parameter set on these (vui.c:202-243): the fields themselves are not used */
    if (br_get1(br)) { if (br_ue(br) > 5 || br_ue(br) > 5) return -1; }                                    /* chroma_sample_loc_type_* */
    if (br_get1(br)) { if (br_get(br, 32) == 0 || br_get(br, 32) == 0) return -1; br_get1(br); }        /* num_units_in_tick, time_scale */
    CHECK(br);
    uint32_t nal_hrd = br_get1(br);
    if (nal_hrd && skip_hrd(br)) return -1;
    uint32_t vcl_hrd = br_get1(br);
    if (vcl_hrd && skip_hrd(br)) return -1;
    if (nal_hrd || vcl_hrd) br_get1(br);                /* low_delay_hrd_flag */
    br_get1(br);                                        /* pic_struct_present */
    s->bitstream_restriction = (uint8_t)br_get1(br);
    if (s->bitstream_restriction) {
        br_get1(br);
        /* max_bytes_per_pic_denom, max_bits_per_mb_denom, log2_max_mv_length_horizontal / _vertical: [0, 16] (vui.c:330-356) */
        for (int i = 0; i < 4; i++) if (br_ue(br) > 16) return -1;
        s->num_reorder_frames = br_ue(br);
        s->max_dec_frame_buffering = br_ue(br);
    }
    CHECK(br);
    return 0;
}

int hd_parse_sps(BitReader *br, Sps *s)
{
    memset(s, 0, sizeof(*s));
    s->profile_idc = (uint8_t)br_get(br, 8);
    s->constraint_flags = (uint8_t)br_get(br, 8);
    s->level_idc = (uint8_t)br_get(br, 8);
    uint32_t id = br_ue(br);
    if (id >= HD_MAX_SPS) return -1;
    s->sps_id = (uint8_t)id;
    uint32_t v = br_ue(br);                              /* log2_max_frame_num_minus4 */
    if (v > 12) return -1;
    s->max_frame_num = 1u << (v + 4);
    v = br_ue(br);
    if (v > 2) return -1;
    s->poc_type = (uint8_t)v;
    if (s->poc_type == 0) {
        v = br_ue(br);
        if (v > 12) return -1;
        s->max_poc_lsb = 1u << (v + 4);
    } else if (s->poc_type == 1) {
        s->delta_pic_order_always_zero = (uint8_t)br_get1(br);
        s->offset_for_non_ref_pic = br_se(br);
        s->offset_for_top_to_bottom_field = br_se(br);
        s->num_ref_frames_in_poc_cycle = br_ue(br);
        if (s->num_ref_frames_in_poc_cycle > 255) return -1;
        for (uint32_t i = 0; i < s->num_ref_frames_in_poc_cycle; i++) s->offset_for_ref_frame[i] = br_se(br);
    }
    s->num_ref_frames = br_ue(br);
    if (s->num_ref_frames > 16) return -1;
    s->gaps_in_frame_num_allowed = (uint8_t)br_get1(br);
    s->width_mbs = br_ue(br) + 1;
    s->height_mbs = br_ue(br) + 1;
    CHECK(br);
    if (!br_get1(br)) return -1;                         /* frame_mbs_only_flag must be 1 */
    br_get1(br);                                         /* direct_8x8_inference_flag */
    s->cropping = (uint8_t)br_get1(br);
    if (s->cropping) {
        s->crop_left = br_ue(br); s->crop_right = br_ue(br);
        s->crop_top = br_ue(br);  s->crop_bottom = br_ue(br);
        CHECK(br);
        if ((int64_t)s->crop_left > 8 * (int64_t)s->width_mbs - ((int64_t)s->crop_right + 1) ||
            (int64_t)s->crop_top > 8 * (int64_t)s->height_mbs - ((int64_t)s->crop_bottom + 1))
            return -1;
    }
    CHECK(br);
    if (s->width_mbs == 0 || s->height_mbs == 0 || s->width_mbs > 1024 || s->height_mbs > 1024) return -1;
    uint32_t pic_mbs = s->width_mbs * s->height_mbs;
    if (pic_mbs > 36864) return -1;
    uint32_t dpb = level_dpb_frames(pic_mbs, s->level_idc);
    if (dpb == 0xFFFFFFFFu || s->num_ref_frames > dpb) dpb = s->num_ref_frames;
    s->max_dpb_size = dpb;
    s->vui_present = (uint8_t)br_get1(br);
    CHECK(br);
    if (s->vui_present) {
        if (parse_vui(br, s)) return -1;
        if (s->bitstream_restriction) {
            if (s->num_reorder_frames > s->max_dec_frame_buffering ||
                s->max_dec_frame_buffering < s->num_ref_frames ||
                s->max_dec_frame_buffering > s->max_dpb_size)
                return -1;
            s->max_dpb_size = s->max_dec_frame_buffering ? s->max_dec_frame_buffering : 1;
        }
    }
    s->valid = 1;
    return 0;   /* trailing-bit damage in parameter sets is tolerated, as in the reference */
}

int hd_sps_equal(const Sps *a, const Sps *b)
{
    if (a->profile_idc != b->profile_idc || a->level_idc != b->level_idc ||
        a->max_frame_num != b->max_frame_num || a->poc_type != b->poc_type ||
        a->num_ref_frames != b->num_ref_frames ||
        a->gaps_in_frame_num_allowed != b->gaps_in_frame_num_allowed ||
        a->width_mbs != b->width_mbs || a->height_mbs != b->height_mbs ||
        a->cropping != b->cropping || a->vui_present != b->vui_present)
        return 0;
    if (a->poc_type == 0 && a->max_poc_lsb != b->max_poc_lsb) return 0;
    if (a->poc_type == 1) {
        if (a->delta_pic_order_always_zero != b->delta_pic_order_always_zero ||
            a->offset_for_non_ref_pic != b->offset_for_non_ref_pic ||
            a->offset_for_top_to_bottom_field != b->offset_for_top_to_bottom_field ||
            a->num_ref_frames_in_poc_cycle != b->num_ref_frames_in_poc_cycle)
            return 0;
        for (uint32_t i = 0; i < a->num_ref_frames_in_poc_cycle; i++)
            if (a->offset_for_ref_frame[i] != b->offset_for_ref_frame[i]) return 0;
    }
    if (a->cropping && (a->crop_left != b->crop_left || a->crop_right != b->crop_right ||
                        a->crop_top != b->crop_top || a->crop_bottom != b->crop_bottom))
        return 0;
    return 1;
}

void hd_free_pps(Pps *p)
{
    if (p) { free(p->slice_group_id); p->slice_group_id = NULL; }
}

int hd_parse_pps(BitReader *br, Pps *p)
{
    memset(p, 0, sizeof(*p));
    uint32_t v = br_ue(br);
    if (v >= HD_MAX_PPS) return -1;
    p->pps_id = (uint8_t)v;
    v = br_ue(br);
    if (v >= HD_MAX_SPS) return -1;
    p->sps_id = (uint8_t)v;
    if (br_get1(br)) return -1;                          /* entropy_coding_mode_flag: CAVLC only */
    p->pic_order_present = (uint8_t)br_get1(br);
    p->num_slice_groups = br_ue(br) + 1;
    if (p->num_slice_groups > 8) return -1;
    if (p->num_slice_groups > 1) {
        v = br_ue(br);
        if (v > 6) return -1;
        p->slice_group_map_type = (uint8_t)v;
        if (v == 0) {
            for (uint32_t i = 0; i < p->num_slice_groups; i++) p->run_length[i] = br_ue(br) + 1;
        } else if (v == 2) {
            for (uint32_t i = 0; i + 1 < p->num_slice_groups; i++) {
                p->top_left[i] = br_ue(br);
                p->bottom_right[i] = br_ue(br);
            }
        } else if (v >= 3 && v <= 5) {
            p->slice_group_change_direction = (uint8_t)br_get1(br);
            p->slice_group_change_rate = br_ue(br) + 1;
        } else if (v == 6) {
            p->pic_size_in_map_units = br_ue(br) + 1;
            CHECK(br);
            if (p->pic_size_in_map_units > 36864) return -1;
            p->slice_group_id = (uint8_t *)malloc(p->pic_size_in_map_units);
            if (!p->slice_group_id) return -2;
            uint32_t bits = p->num_slice_groups > 4 ? 3 : p->num_slice_groups > 2 ? 2 : 1;
            for (uint32_t i = 0; i < p->pic_size_in_map_units; i++) {
                uint32_t g = br_get(br, bits);
                if (g >= p->num_slice_groups) { hd_free_pps(p); return -1; }
                p->slice_group_id[i] = (uint8_t)g;
            }
        }
    }
    p->num_ref_idx_l0_active = br_ue(br) + 1;
    if (p->num_ref_idx_l0_active > 32) goto bad;
    if (br_ue(br) > 31) goto bad;                        /* num_ref_idx_l1_active_minus1 */
    if (br_get1(br)) goto bad;                           /* weighted_pred_flag */
    if (br_get(br, 2) > 2) goto bad;                     /* weighted_bipred_idc */
    {
        int32_t q = br_se(br);
        if (q < -26 || q > 25) goto bad;
        p->pic_init_qp = 26 + q;
        q = br_se(br);                                   /* pic_init_qs */
        if (q < -26 || q > 25) goto bad;
        q = br_se(br);
        if (q < -12 || q > 12) goto bad;
        p->chroma_qp_index_offset = q;
    }
    p->deblocking_filter_control_present = (uint8_t)br_get1(br);
    p->constrained_intra_pred = (uint8_t)br_get1(br);
    p->redundant_pic_cnt_present = (uint8_t)br_get1(br);
    if (br_overrun(br)) goto bad;
    p->valid = 1;
    return 0;
bad:
    hd_free_pps(p);
    return -1;
}
