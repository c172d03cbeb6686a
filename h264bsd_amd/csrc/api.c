/*
 * api.c — the exported h264bsd* C API (include/h264bsd_decoder.h) on top of the host parser and
 * the HIP engine.  Mirrors the reference's public functions one to one
 * (src/h264bsd_decoder.c:90-1370); see the header for the per-function citations.
 *
 * There is no CPU pixel path in this library: h264bsdInit() binds the instance to the HIP engine
 * and fails loudly (stderr + HANTRO_NOK) when no gfx950 device is usable.  The only GPU-less mode is
 * h264bsdmiInitCapture(), which produces frame jobs and never pixels.
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <sched.h>
#include <stdatomic.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include "../../include/h264bsd_mi355x_bench.h"
#include "hostdec.h"
#include "engine.h"

HostDec *hd_create(int no_output_reordering);
void hd_destroy(HostDec *d);
int hd_decode(HostDec *d, uint8_t *stream, uint32_t len, uint32_t pic_id, uint32_t *read_bytes);

typedef struct ApiDec {
    HostDec *hd;
    h264bsdmi_job_cb cb;
    void *cb_user;
} ApiDec;

/* Every H264BSDMI_* variable this library reads (tests/test_abi.py compares the list with the sources and INTEGRATION.md).  A
 * variable of that family that is NOT in the list — a typing error, or a switch of an earlier release — is named on stderr when
 * the library is loaded instead of being silently ignored. */
extern char **environ;
__attribute__((constructor)) static void report_unknown_switches(void)
{
    static const char *const known[] = { "H264BSDMI_LANES", "H264BSDMI_TAIL", "H264BSDMI_BAND_BUDGET", "H264BSDMI_HEAVY_BUDGET", "H264BSDMI_TRACE_LANES",
                                         "H264BSDMI_COPY_ELISION", "H264BSDMI_THREADS", "H264BSDMI_HOST_SHARE", "H264BSDMI_PIN", NULL };
    for (char **e = environ; e && *e; e++) {
        if (strncmp(*e, "H264BSDMI_", 10) != 0) continue;
        const char *eq = strchr(*e, '=');
        const size_t len = eq ? (size_t)(eq - *e) : strlen(*e);
        int ok = 0;
        for (int k = 0; known[k] && !ok; k++) ok = strlen(known[k]) == len && strncmp(known[k], *e, len) == 0;
        if (!ok) fprintf(stderr, "h264bsd-mi355x: unknown environment switch %.*s ignored (INTEGRATION.md lists the ones this library reads)\n", (int)len, *e);
    }
}

static ApiDec *dec_of(storage_t *s) { return s ? (ApiDec *)s->opaque : NULL; }

/* ---- capture sink ---- */
static int cap_configure(void *u, uint32_t w, uint32_t h, uint32_t n) { (void)u; (void)w; (void)h; (void)n; return 0; }
static int cap_submit(void *u, const uint8_t *blob, uint32_t bytes)
{
    ApiDec *a = (ApiDec *)u;
    if (a->cb) a->cb(a->cb_user, blob, bytes);
    return 0;
}

static u32 init_common(storage_t *s, u32 no_reorder, h264bsdmi_job_cb cb, void *user, int capture)
{
    if (!s) return HANTRO_NOK;
    memset(s, 0, sizeof(*s));
    ApiDec *a = (ApiDec *)calloc(1, sizeof(ApiDec));
    if (!a) return HANTRO_NOK;
    a->hd = hd_create((int)no_reorder);
    if (!a->hd) { free(a); return HANTRO_NOK; }
    if (capture) {
        a->cb = cb;
        a->cb_user = user;
        a->hd->sink.user = a;
        a->hd->sink.configure = cap_configure;
        a->hd->sink.submit = cap_submit;
    } else if (eng_attach(&a->hd->sink)) {
        fprintf(stderr, "h264bsd-mi355x: h264bsdInit failed: no usable HIP device (this library has no CPU pixel path)\n");
        hd_destroy(a->hd);
        free(a);
        return HANTRO_NOK;
    } else {
        /* bound to a device: the frames live from job to job, copies that would rewrite what a tile holds are left out
         * (hostdec.h, copy elision).  H264BSDMI_COPY_ELISION=0 switches it off for the process. */
        const char *e = getenv("H264BSDMI_COPY_ELISION");
        a->hd->copy_elision = !(e && *e == '0');
        a->hd->sink_errors_at_start = a->hd->sink.errors ? a->hd->sink.errors(a->hd->sink.user) : 0;
    }
    s->opaque = a;
    return HANTRO_OK;
}

u32 h264bsdInit(storage_t *s, u32 noOutputReordering) { return init_common(s, noOutputReordering, NULL, NULL, 0); }
u32 h264bsdmiInitCapture(storage_t *s, u32 noOutputReordering, h264bsdmi_job_cb cb, void *user)
{
    return init_common(s, noOutputReordering, cb, user, 1);
}

void h264bsdShutdown(storage_t *s)
{
    ApiDec *a = dec_of(s);
    if (!a) return;
    hd_destroy(a->hd);
    free(a);
    s->opaque = NULL;
}

storage_t *h264bsdAlloc(void) { return (storage_t *)calloc(1, sizeof(storage_t)); }
void h264bsdFree(storage_t *s) { free(s); }

u32 h264bsdDecode(storage_t *s, u8 *byteStrm, u32 len, u32 picId, u32 *readBytes)
{
    ApiDec *a = dec_of(s);
    if (!a || !byteStrm || !len || !readBytes) return H264BSD_ERROR;
    return (u32)hd_decode(a->hd, byteStrm, len, picId, readBytes);
}

static const OutPic *pop_output(ApiDec *a, u32 *picId, u32 *isIdrPic, u32 *numErrMbs)
{
    const OutPic *o = hd_dpb_next_output(&a->hd->dpb);
    if (!o) return NULL;
    if (picId) *picId = o->pic_id;
    if (isIdrPic) *isIdrPic = o->is_idr;
    if (numErrMbs) *numErrMbs = o->num_err_mbs;
    return o;
}

u8 *h264bsdNextOutputPicture(storage_t *s, u32 *picId, u32 *isIdrPic, u32 *numErrMbs)
{
    ApiDec *a = dec_of(s);
    if (!a) return NULL;
    const OutPic *o = pop_output(a, picId, isIdrPic, numErrMbs);
    if (!o || !a->hd->sink.fetch) return NULL;
    return a->hd->sink.fetch(a->hd->sink.user, o->slot);
}

int h264bsdmiSetInputReadOnly(storage_t *s, u32 on)
{
    ApiDec *a = dec_of(s);
    if (!a) return -1;
    a->hd->input_readonly = on ? 1 : 0;
    return 0;
}

int h264bsdmiSetCopyElision(storage_t *s, u32 on)
{
    ApiDec *a = dec_of(s);
    if (!a) return -1;
    a->hd->copy_elision = on ? 1 : 0;
    return 0;
}

int h264bsdmiNextOutputInfo(storage_t *s, u32 *picId, u32 *isIdrPic, u32 *numErrMbs)
{
    ApiDec *a = dec_of(s);
    if (!a) return -1;
    const OutPic *o = pop_output(a, picId, isIdrPic, numErrMbs);
    return o ? (int)o->slot : -1;
}

int h264bsdmiNextOutputPictureDevice(storage_t *s, int format, int crop, h264bsdmi_device_picture *out)
{
    ApiDec *a = dec_of(s);
    if (!a || !out || format < 0 || format > 3) return -1;
    if (!a->hd->sink.fetch_device) return -1;               /* capture mode: there are no pixels */
    const Sps *sps = a->hd->active_sps;
    if (!sps) return 0;
    u32 id = 0, idr = 0, nerr = 0;
    const OutPic *o = pop_output(a, &id, &idr, &nerr);
    if (!o) return 0;
    u32 x0 = 0, y0 = 0, w = 16 * sps->width_mbs, h = 16 * sps->height_mbs;
    if (crop && sps->cropping) {
        x0 = 2 * sps->crop_left;
        y0 = 2 * sps->crop_top;
        w -= 2 * (sps->crop_left + sps->crop_right);
        h -= 2 * (sps->crop_top + sps->crop_bottom);
    }
    void *stream = NULL;
    void *p = a->hd->sink.fetch_device(a->hd->sink.user, o->slot, format, x0, y0, w, h, &stream);
    if (!p) return -2;
    out->data = p;
    out->width = w;
    out->height = h;
    out->pitch = format == H264BSDMI_FMT_I420 ? w : 4 * w;
    out->format = (u32)format;
    out->picId = id;
    out->isIdrPic = idr;
    out->numErrMbs = nerr;
    out->stream = stream;
    return 1;
}

static u32 *next_converted(storage_t *s, u32 *picId, u32 *isIdrPic, u32 *numErrMbs, int fmt)
{
    ApiDec *a = dec_of(s);
    if (!a) return NULL;
    const OutPic *o = pop_output(a, picId, isIdrPic, numErrMbs);
    if (!o || !a->hd->sink.fetch_converted) return NULL;
    return a->hd->sink.fetch_converted(a->hd->sink.user, o->slot, fmt);
}
u32 *h264bsdNextOutputPictureRGBA(storage_t *s, u32 *p, u32 *i, u32 *n) { return next_converted(s, p, i, n, 0); }
u32 *h264bsdNextOutputPictureBGRA(storage_t *s, u32 *p, u32 *i, u32 *n) { return next_converted(s, p, i, n, 1); }
u32 *h264bsdNextOutputPictureYCbCrA(storage_t *s, u32 *p, u32 *i, u32 *n) { return next_converted(s, p, i, n, 2); }

static const Sps *active_sps(storage_t *s)
{
    ApiDec *a = dec_of(s);
    return a ? a->hd->active_sps : NULL;
}

u32 h264bsdPicWidth(storage_t *s) { const Sps *p = active_sps(s); return p ? p->width_mbs : 0; }
u32 h264bsdPicHeight(storage_t *s) { const Sps *p = active_sps(s); return p ? p->height_mbs : 0; }
u32 h264bsdProfile(storage_t *s) { const Sps *p = active_sps(s); return p ? p->profile_idc : 0; }

u32 h264bsdVideoRange(storage_t *s)
{
    const Sps *p = active_sps(s);
    return (p && p->vui_present && p->video_signal_type_present && p->video_full_range) ? 1 : 0;
}
u32 h264bsdMatrixCoefficients(storage_t *s)
{
    const Sps *p = active_sps(s);
    if (p && p->vui_present && p->video_signal_type_present && p->colour_description_present)
        return p->matrix_coefficients;
    return 2;   /* unspecified */
}

void h264bsdCroppingParams(storage_t *s, u32 *croppingFlag, u32 *left, u32 *width, u32 *top, u32 *height)
{
    const Sps *p = active_sps(s);
    if (p && p->cropping) {
        *croppingFlag = 1;
        *left = 2 * p->crop_left;
        *width = 16 * p->width_mbs - 2 * (p->crop_left + p->crop_right);
        *top = 2 * p->crop_top;
        *height = 16 * p->height_mbs - 2 * (p->crop_top + p->crop_bottom);
    } else {
        *croppingFlag = 0;
        *left = *width = *top = *height = 0;
    }
}

void h264bsdSampleAspectRatio(storage_t *s, u32 *sarWidth, u32 *sarHeight)
{
    /* Table E-1 */
    static const u8 sar[14][2] = { { 0, 0 }, { 1, 1 }, { 12, 11 }, { 10, 11 }, { 16, 11 }, { 40, 33 }, { 24, 11 },
        { 20, 11 }, { 32, 11 }, { 80, 33 }, { 18, 11 }, { 15, 11 }, { 64, 33 }, { 160, 99 } };
    const Sps *p = active_sps(s);
    u32 w = 1, h = 1;
    if (p && p->vui_present && p->aspect_ratio_present) {
        if (p->aspect_ratio_idc < 14) { w = sar[p->aspect_ratio_idc][0]; h = sar[p->aspect_ratio_idc][1]; }
        else if (p->aspect_ratio_idc == 255) {
            w = p->sar_width; h = p->sar_height;
            if (!w || !h) w = h = 0;
        } else w = h = 0;
    }
    *sarWidth = w;
    *sarHeight = h;
}

/* 1 when at least one PPS with its SPS is stored and consistent (reference storage.c:793-829) */
u32 h264bsdCheckValidParamSets(storage_t *s)
{
    ApiDec *a = dec_of(s);
    if (!a) return 0;
    for (int i = 0; i < HD_MAX_PPS; i++) {
        const Pps *p = a->hd->pps[i];
        if (p && a->hd->sps[p->sps_id] && !hd_check_pps(p, a->hd->sps[p->sps_id])) return 1;     /* h264bsdValidParamSets, storage.c:863-885 */
    }
    return 0;
}

void h264bsdFlushBuffer(storage_t *s)
{
    ApiDec *a = dec_of(s);
    if (a) hd_dpb_flush(&a->hd->dpb);
}

/* test/tooling hook: complete a hand-built frame job (see fj_finalize, hd_core.c) */
int h264bsdmiJobFinalize(u8 *job, u32 capacity, u32 n_coef_blocks)
{
    /* a hand-built job carries no FJ_CODED_WIDE flags (the parser sets them while it sums up the levels): derive them from
     * the coefficient blocks, with the parser's own bound */
    FjHeader *h = (FjHeader *)job;
    if (capacity >= sizeof(FjHeader) && (size_t)h->rec_off + (size_t)h->n_mbs * sizeof(FjMbRec) <= capacity) {
        FjMbRec *recs = (FjMbRec *)(job + h->rec_off);
        for (u32 a = 0; a < h->n_mbs; a++) {
            FjMbRec *r = &recs[a];
            r->coded &= ~FJ_CODED_WIDE;
            if (r->kind != FJ_MB_INTER || !(r->coded & 0x02FFFFFFu)) continue;
            const u32 n_l = (u32)__builtin_popcount(r->coded & 0xFFFFu), n_c = (u32)__builtin_popcount((r->coded >> 16) & 0xFFu);
            const u32 has_cdc = (r->coded & FJ_CODED_CHROMA_DC) ? 1u : 0u;
            if ((size_t)h->coef_off + ((size_t)r->coef_idx + n_l + has_cdc + n_c) * 32u > capacity) continue;      /* (fj_finalize rejects the job) */
            const int16_t *p = (const int16_t *)(job + h->coef_off) + 16u * (size_t)r->coef_idx;
            u32 sl = 0, sd = 0, sc = 0;                 /* the largest block's sum of each kind, like the parser's (hd_mb.c parse_residual) */
            for (u32 b = 0; b < n_l; b++, p += 16) { u32 t = 0; for (int i = 0; i < 16; i++) t += (u32)abs(p[i]); if (t > sl) sl = t; }
            if (has_cdc) {
                u32 t0 = 0, t1 = 0;
                for (int i = 0; i < 4; i++) { t0 += (u32)abs(p[i]); t1 += (u32)abs(p[4 + i]); }
                sd = t0 > t1 ? t0 : t1;
                p += 16;
            }
            for (u32 b = 0; b < n_c; b++, p += 16) { u32 t = 0; for (int i = 0; i < 16; i++) t += (u32)abs(p[i]); if (t > sc) sc = t; }
            if (!hd_residual_bound_ok(sl, sd, sc, r->qp_y, r->qp_c)) r->coded |= FJ_CODED_WIDE;
        }
    }
    return fj_finalize(job, capacity, n_coef_blocks);
}

void h264bsdConvertToRGBA(u32 width, u32 height, u8 *data, u32 *pOutput) { eng_convert_host(0, width, height, data, pOutput); }
void h264bsdConvertToBGRA(u32 width, u32 height, u8 *data, u32 *pOutput) { eng_convert_host(1, width, height, data, pOutput); }
void h264bsdConvertToYCbCrA(u32 width, u32 height, u8 *data, u32 *pOutput) { eng_convert_host(2, width, height, data, pOutput); }

/* ================================================================== host parse pipeline */
u32 h264bsdmiDecodePicture(storage_t *s, u8 *buf, u32 len, u32 picId, u32 *consumed, u32 *nErrors)
{
    u32 off = 0, last = H264BSD_RDY, errs = 0, stalls = 0;
    while (off < len) {
        u32 rb = 0;
        last = h264bsdDecode(s, buf + off, len - off, picId, &rb);
        off += rb;
        if (last == H264BSD_PIC_RDY) break;
        if (last >= H264BSD_ERROR) errs++;
        stalls = rb ? 0 : stalls + 1;              /* HDRS_RDY and friends repeat the same NAL once; never spin */
        if (stalls > 3) break;
    }
    if (consumed) *consumed = off;
    if (nErrors) *nErrors = errs;
    return last == H264BSD_PIC_RDY ? last : (u32)H264BSD_RDY;
}

/* A small persistent pool: workers sleep on a condition variable, a batch is an index range handed out by an atomic
 * counter (pictures differ a lot in parse time: I pictures cost several times a P picture).
 * One process may drive several GPUs (h264bsdmiSetDevice per thread): the workers are then split into one group per
 * device in use, a worker first takes the pictures of decoder instances that live on ITS device, and (H264BSDMI_PIN=2,
 * the default) is pinned to the CPUs of that device's NUMA node (eng_device_cpus: /sys/bus/pci/devices/<gpu>/
 * local_cpulist), so that the frame job it builds in pinned staging memory and the parser state it touches sit next to
 * the GPU they feed.  H264BSDMI_PIN=1: spread over all CPUs; 0: no pinning. */
typedef struct Batch {
    u32 n;
    storage_t *const *dec;
    u8 *const *buf;
    const u32 *len, *pic_id;
    u32 *status, *consumed, *n_errors;
    u8 **out;                               /* non-NULL: item i pulls first — h264bsdNextOutputPicture(dec[i]) -> out[i], out_id[i], out_idr[i], out_nerr[i] */
    u32 *out_id, *out_idr, *out_nerr;       /* (each may be NULL) */
    atomic_uchar *taken;                    /* one flag per item */
    unsigned char *began;                   /* per item, touched by the worker that took it only: its picture is on its way (item_pull_begin) */
    int active;                             /* pool threads currently inside this batch (guarded by g_pool.mu) */
    signed char *item_dev;                  /* device of every item's decoder instance (-1: capture mode) */
    int devs[16], n_devs;                   /* the devices in use in this batch */
    u32 workers;                            /* threads that work on this batch (run_batch): all of the pool, or fewer when the batch pulls */
    atomic_int pull_failed;                 /* a picture was popped from an instance's output queue and could not be fetched (HIP failure): the batch returns -1 */
} Batch;

static struct {
    pthread_mutex_t mu;
    pthread_cond_t wake, idle;
    pthread_t th[64];
    int n_threads, started, want, pin;
    int pull_workers;                       /* threads a batch that pulls pictures runs on (run_batch): the CPUs the process may have running at once — or
                                               all of the pool when the application (or H264BSDMI_THREADS) named the number of threads */
    unsigned generation;
    Batch *batch;
    pthread_mutex_t api_mu;                 /* one batch at a time */
} g_pool = { PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER, { 0 }, 0, 0, 0, 0, 0, 0, NULL,
             PTHREAD_MUTEX_INITIALIZER };

/* An item of a batch that pulls (b->out) runs in two steps: item_pull_begin pops the instance's output queue and starts the picture on
 * its way to host memory, item_finish waits for it and — if the batch decodes as well (b->buf) — parses the instance's next picture.  A
 * worker begins its NEXT item before it finishes the current one (batch_take): while it parses, the next picture crosses the link, and
 * its wait finds the picture there (8.5 -> 8.8 k fps with 256 instances and 20 threads; DESIGN.md §5 has the whole account of the
 * host-output path, including what is still unexplained about it: a round gets LONGER with more pool threads). */
static void item_pull_begin(Batch *b, u32 i)
{
    ApiDec *a = dec_of(b->dec[i]);
    u32 id = 0, idr = 0, nerr = 0;
    const OutPic *o = a ? pop_output(a, &id, &idr, &nerr) : NULL;
    b->out[i] = NULL;
    if (b->out_id) b->out_id[i] = id;
    if (b->out_idr) b->out_idr[i] = idr;
    if (b->out_nerr) b->out_nerr[i] = nerr;
    if (!o) return;
    const JobSink *k = &a->hd->sink;
    if (k->fetch_begin && k->fetch_end) {
        if (k->fetch_begin(k->user, o->slot) == 0) b->began[i] = 1;        /* on its way: item_finish collects it */
        else atomic_store(&b->pull_failed, 1);
    } else if (k->fetch) {
        b->out[i] = k->fetch(k->user, o->slot);
        if (!b->out[i]) atomic_store(&b->pull_failed, 1);
    }
}

static void item_finish(Batch *b, u32 i)
{
    if (b->out) {
        const ApiDec *a = dec_of(b->dec[i]);
        if (b->began[i]) {
            b->out[i] = a->hd->sink.fetch_end(a->hd->sink.user);
            if (!b->out[i]) atomic_store(&b->pull_failed, 1);
        }
        if (!b->buf) return;
        /* pull AND decode: the reference's decoder drops what is still waiting in its output queue when the next slice arrives
         * (src/h264bsd_dpb.c:1260-1261), so an instance with more pictures to give is not fed — the caller comes back for them */
        if (a && a->hd->dpb.out_idx < a->hd->dpb.n_out) {
            b->status[i] = H264BSD_RDY; b->consumed[i] = 0;
            if (b->n_errors) b->n_errors[i] = 0;
            return;
        }
    }
    b->status[i] = h264bsdmiDecodePicture(b->dec[i], b->buf[i], b->len[i], b->pic_id ? b->pic_id[i] : 0,
                                          &b->consumed[i], b->n_errors ? &b->n_errors[i] : NULL);
}

/* item i has just been claimed by this worker; *held = the item it began before and has not finished (-1: none) */
static void batch_take(Batch *b, u32 i, int *held)
{
    if (!b->out) { item_finish(b, i); return; }
    item_pull_begin(b, i);
    if (*held >= 0) item_finish(b, (u32)*held);
    *held = (int)i;
}

/* Worker `me` of `nw`: group g = me % n_devs serves device devs[g].  It takes the items of its device i = k, k + gs,
 * ... (k = its index in the group, gs = group size) first — so that a stream is normally parsed by the same thread,
 * from the same NUMA node, every round — then whatever else is left anywhere. */
static void batch_work(Batch *b, u32 me, u32 nw)
{
    const u32 nd = b->n_devs > 0 ? (u32)b->n_devs : 1u;
    const u32 g = me % nd, k = me / nd, gs = (nw - g + nd - 1) / nd;
    const int my_dev = b->n_devs > 0 ? b->devs[g] : -1;
    u32 seen = 0;
    int held = -1;
    for (u32 i = 0; i < b->n; i++) {
        if (b->n_devs > 0 && b->item_dev[i] != my_dev) continue;
        if (seen++ % gs == k && !atomic_exchange(&b->taken[i], 1)) batch_take(b, i, &held);
    }
    for (int pass = 0; pass < 2; pass++)        /* leftovers: own device first */
        for (u32 i = 0; i < b->n; i++) {
            if (pass == 0 && b->n_devs > 0 && b->item_dev[i] != my_dev) continue;
            if (!atomic_load(&b->taken[i]) && !atomic_exchange(&b->taken[i], 1)) batch_take(b, i, &held);
        }
    if (held >= 0) item_finish(b, (u32)held);
}

static void pin_worker(u32 me, int device, int *pinned_to)
{
    if (g_pool.pin == 1 && *pinned_to != -2) {
        /* spread the workers over the CPUs (worker k -> CPU k * ncpu / nthreads) */
        const long ncpu = sysconf(_SC_NPROCESSORS_ONLN);
        cpu_set_t set;
        CPU_ZERO(&set);
        CPU_SET((int)(((long)me * ncpu) / (g_pool.want > 0 ? g_pool.want : 1)) % (int)ncpu, &set);
        pthread_setaffinity_np(pthread_self(), sizeof(set), &set);
        *pinned_to = -2;
    } else if (g_pool.pin == 2 && device >= 0 && *pinned_to != device) {
        int cpus[1024];
        const int n = eng_device_cpus(device, cpus, 1024);
        if (n > 0) {
            /* only CPUs this process may run on (a cpuset-restricted container may not own the GPU's node at all: then the
             * worker stays where the scheduler puts it) */
            cpu_set_t set, allowed;
            CPU_ZERO(&set);
            int usable = 0;
            if (sched_getaffinity(0, sizeof(allowed), &allowed) == 0) {
                for (int i = 0; i < n; i++) if (cpus[i] < CPU_SETSIZE && CPU_ISSET(cpus[i], &allowed)) { CPU_SET(cpus[i], &set); usable++; }
            }
            if (usable > 0) pthread_setaffinity_np(pthread_self(), sizeof(set), &set);
        }
        *pinned_to = device;
    }
}

static void *pool_main(void *arg)
{
    const u32 me = (u32)(size_t)arg;        /* 1.. ; the caller's thread is worker 0 (never pinned: it is the application's) */
    int pinned_to = -1;
    unsigned seen = 0;
    pthread_mutex_lock(&g_pool.mu);
    for (;;) {
        while (g_pool.generation == seen) pthread_cond_wait(&g_pool.wake, &g_pool.mu);
        seen = g_pool.generation;
        Batch *b = g_pool.batch;
        if (!b) continue;
        b->active++;
        const u32 nw = b->workers;
        pthread_mutex_unlock(&g_pool.mu);
        if (me < nw) {
            pin_worker(me, b->n_devs > 0 ? b->devs[me % (u32)b->n_devs] : -1, &pinned_to);
            batch_work(b, me, nw);
        }
        pthread_mutex_lock(&g_pool.mu);
        if (--b->active == 0) pthread_cond_broadcast(&g_pool.idle);
    }
    return NULL;
}

/* CPUs this process may really use: the online count, cut down by the affinity mask and by a cgroup CPU quota
 * (containers: /sys/fs/cgroup/cpu.max, or cpu.cfs_quota_us / cpu.cfs_period_us under cgroup v1).  Parser threads
 * beyond that only take turns: on a box with 256 hardware threads and a quota of 16 CPUs, 64 threads parse 24 % fewer
 * pictures per second than 16. */
static long usable_cpus_q(int quota_bonus)
{
    long n = sysconf(_SC_NPROCESSORS_ONLN);
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0 && CPU_COUNT(&set) > 0 && CPU_COUNT(&set) < n) n = CPU_COUNT(&set);
    long long quota = -1, period = 0;
    FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r");
    if (f) {
        char q[32] = "";
        if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atoll(q);
        fclose(f);
    } else if ((f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r"))) {
        if (fscanf(f, "%lld", &quota) != 1) quota = -1;
        fclose(f);
        if ((f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r"))) { if (fscanf(f, "%lld", &period) != 1) period = 0; fclose(f); }
    }
    if (quota > 0 && period > 0) {
        /* A quota is CPU time per period, not a set of cores: with exactly `quota` threads the budget is not used up (the
         * thread that enqueues the device work between two rounds is not parsing, threads wait for each other at the end
         * of a round).  A quarter more threads than CPUs of quota measured best on the GPU box (256 hardware threads, quota
         * 16: 16 / 20 / 24 / 28 / 32 threads -> 17.5 / 18.7 / 16.8 / 16.9 / 16.0 k pictures per second end to end). */
        const long q = (long)((quota + period - 1) / period), q125 = q + q / 4;
        if (q >= 1 && q < n) n = !quota_bonus ? q : q125 < n ? q125 : n;
    }
    return n < 1 ? 1 : n;
}
static long usable_cpus(void) { return usable_cpus_q(1); }

/* Processes that share this host's CPUs, one per GPU (the usual multi-GPU launch): every one of them sizes its
 * parser pool for ITS share of the CPUs the container may use — eight ranks that each start usable_cpus() threads only take
 * turns.  H264BSDMI_HOST_SHARE says how many; without it the launcher's LOCAL_WORLD_SIZE (torch.distributed.run, mpirun
 * wrappers) is taken. */
static long host_share(void)
{
    const char *e = getenv("H264BSDMI_HOST_SHARE");
    int from_launcher = 0;
    if (!e || !*e) {
        /* The launcher's LOCAL_WORLD_SIZE only says that N processes run on this host — not that they share CPUs.  Where the launcher
         * (SLURM / mpirun binding, numactl or a cgroup per rank) has already given this process its own CPUs, usable_cpus() IS the
         * share, and dividing again would leave a CPU-bound pipeline with 1 / N of its threads.  The fallback therefore applies
         * only while the process's affinity mask is the whole host.  A cgroup CPU quota does not count as a partition:
         * a container's quota is normally shared by all the ranks started inside it (the GPU boxes of this project: 256 hardware
         * threads online, quota 16 for the container) — a launcher that gives every rank its own cgroup sets H264BSDMI_HOST_SHARE=1. */
        e = getenv("LOCAL_WORLD_SIZE");
        const long online = sysconf(_SC_NPROCESSORS_ONLN);
        cpu_set_t set;
        if (e && online > 0 && sched_getaffinity(0, sizeof(set), &set) == 0 && CPU_COUNT(&set) > 0 && CPU_COUNT(&set) < online) e = NULL;
        from_launcher = e != NULL;
    }
    long n = e ? atol(e) : 1;
    n = n < 1 ? 1 : n > 64 ? 64 : n;
    static int said;
    if (from_launcher && n > 1 && !said) {
        said = 1;
        fprintf(stderr, "h264bsd-mi355x: parser pool sized for 1 / %ld of this host's CPUs (LOCAL_WORLD_SIZE=%ld; H264BSDMI_HOST_SHARE or H264BSDMI_THREADS override)\n", n, n);
    }
    return n;
}

static int pool_default_threads(void)
{
    const char *e = getenv("H264BSDMI_THREADS");
    long n = e ? atol(e) : (usable_cpus() + host_share() - 1) / host_share();
    if (n < 1) n = 1;
    if (n > 64) n = 64;
    return (int)n;
}

int h264bsdmiSetParserThreads(int n)
{
    pthread_mutex_lock(&g_pool.mu);
    const int chosen = n >= 1 || getenv("H264BSDMI_THREADS") != NULL;
    if (n < 1) n = pool_default_threads();
    if (n > 64) n = 64;
    g_pool.want = n;
    g_pool.pin = getenv("H264BSDMI_PIN") ? atoi(getenv("H264BSDMI_PIN")) : 2;
    /* the caller's thread works too: n threads in total = n-1 pool threads (threads are only ever added) */
    while (g_pool.started < n - 1) {
        if (pthread_create(&g_pool.th[g_pool.started], NULL, pool_main, (void *)(size_t)(g_pool.started + 1))) break;
        pthread_detach(g_pool.th[g_pool.started]);
        g_pool.started++;
    }
    g_pool.n_threads = g_pool.started + 1;
    {
        const long lim = (usable_cpus_q(0) + host_share() - 1) / host_share();
        g_pool.pull_workers = chosen || lim < 1 || lim > g_pool.n_threads ? g_pool.n_threads : (int)lim;
    }
    const int r = g_pool.n_threads;
    pthread_mutex_unlock(&g_pool.mu);
    return r;
}

static int run_batch(Batch *b)
{
    const u32 n = b->n;
    pthread_mutex_lock(&g_pool.api_mu);
    if (!g_pool.n_threads) h264bsdmiSetParserThreads(0);
    atomic_uchar *taken = (atomic_uchar *)calloc(n, sizeof(atomic_uchar));
    unsigned char *began = (unsigned char *)calloc(n, 1);
    signed char *item_dev = (signed char *)malloc(n);
    if (!taken || !began || !item_dev) { free(taken); free(began); free(item_dev); pthread_mutex_unlock(&g_pool.api_mu); return -1; }
    b->taken = taken; b->began = began; b->item_dev = item_dev; b->active = 0; b->n_devs = 0;
    atomic_store(&b->pull_failed, 0);
    for (u32 i = 0; i < n; i++) {
        const ApiDec *a = dec_of(b->dec[i]);
        const int dv = a && a->hd ? eng_sink_device(&a->hd->sink) : -1;
        item_dev[i] = (signed char)dv;
        int known = dv < 0;
        for (int k = 0; k < b->n_devs && !known; k++) known = b->devs[k] == dv;
        if (!known && b->n_devs < 16) b->devs[b->n_devs++] = dv;
    }
    pthread_mutex_lock(&g_pool.mu);
    /* A batch that pulls pictures runs on no more threads than the process may have running at once.  The pool is a quarter larger than a
     * CPU quota (usable_cpus_q: that uses the budget up when threads only parse); threads that also wait for the device in between burn a
     * period's budget in bursts, and while the cgroup is throttled nothing feeds the link: 256 pulls + parses take 26.5 ms on 16 threads,
     * 29.5 on 20, 52 on 40 under a quota of 16 CPUs. */
    const u32 nw = (u32)(b->out ? g_pool.pull_workers : g_pool.n_threads);
    b->workers = nw;
    g_pool.batch = b;
    g_pool.generation++;
    pthread_cond_broadcast(&g_pool.wake);
    pthread_mutex_unlock(&g_pool.mu);
    batch_work(b, 0, nw);
    pthread_mutex_lock(&g_pool.mu);
    g_pool.batch = NULL;                    /* late wakers find nothing; those inside are counted in b.active */
    while (b->active) pthread_cond_wait(&g_pool.idle, &g_pool.mu);
    pthread_mutex_unlock(&g_pool.mu);
    free(taken);
    free(began);
    free(item_dev);
    pthread_mutex_unlock(&g_pool.api_mu);
    return atomic_load(&b->pull_failed) ? -1 : 0;      /* -1: a picture left an output queue and could not be fetched (its out[] entry is NULL) */
}

int h264bsdmiDecodePictureBatch(u32 n, storage_t *const *dec, u8 *const *buf, const u32 *len, const u32 *picId,
                                u32 *status, u32 *consumed, u32 *nErrors)
{
    if (!dec || !buf || !len || !status || !consumed) return -1;
    if (!n) return 0;
    Batch b = { n, dec, buf, len, picId, status, consumed, nErrors, NULL, NULL, NULL, NULL, NULL, NULL, 0, NULL, { 0 }, 0, 0, 0 };
    return run_batch(&b);
}

/* h264bsdNextOutputPicture() of n decoder instances at once, on the parser pool's threads: every thread enqueues its pictures'
 * way out (layout kernel + copy to pinned host memory) and waits for ITS copies only, so the transfers of the instances overlap
 * each other and the device's remaining work.  pictures[i] = NULL when instance i has no picture to give. */
int h264bsdmiNextOutputPictureBatch(u32 n, storage_t *const *dec, u8 **pictures, u32 *picId, u32 *isIdrPic, u32 *numErrMbs)
{
    if (!dec || !pictures) return -1;
    if (!n) return 0;
    Batch b = { n, dec, NULL, NULL, NULL, NULL, NULL, NULL, pictures, picId, isIdrPic, numErrMbs, NULL, NULL, 0, NULL, { 0 }, 0, 0, 0 };
    return run_batch(&b);
}

/* One round of the reference's per-stream loop (posix/test_h264bsd.c:146-177: pull what is ready, then decode on) for n instances
 * at once: every pool thread pulls ITS instance's next picture and then parses that instance's next picture, so the transfers of
 * some instances run beside the parsing of others — with h264bsdmiNextOutputPictureBatch + h264bsdmiDecodePictureBatch the
 * CPUs idle while the pictures cross the link and the link idles while the CPUs parse. */
int h264bsdmiPullAndDecodePictureBatch(u32 n, storage_t *const *dec, u8 **pictures, u32 *outPicId, u32 *outIsIdrPic, u32 *outNumErrMbs,
                                       u8 *const *buf, const u32 *len, const u32 *picId, u32 *status, u32 *consumed, u32 *nErrors)
{
    if (!dec || !pictures || !buf || !len || !status || !consumed) return -1;
    if (!n) return 0;
    Batch b = { n, dec, buf, len, picId, status, consumed, nErrors, pictures, outPicId, outIsIdrPic, outNumErrMbs, NULL, NULL, 0, NULL, { 0 }, 0, 0, 0 };
    return run_batch(&b);
}
