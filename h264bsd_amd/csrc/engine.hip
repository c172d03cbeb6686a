/*
 * engine.hip — host side of the HIP engine: device context, per-decoder DPB in HBM (macroblock tiles,
 * kernels.hip.h), lazy batched execution of queued frame jobs ("ticks"), and the HBM-resident replay sets used by
 * bench.py and the kernel tests (exported by the bench library only, include/h264bsd_mi355x_bench.h).
 *
 * Execution model.  A tick holds at most one picture per stream (pictures of one stream depend on each other
 * through the DPB; pictures of different streams never do).  A tick of N pictures is SIX launches (launch_tick):
 *     k_copy            grid (32, N)          x 256  whole-sample copy macroblocks, one run of <= 8 tiles per wavefront; on a HIP stream of its own
 *                                                    next to the inter kernels (single-lane engine and replay sets: SideLane), joined before k_frame_intra
 *     k_recon_inter<0>  grid (n_uni/4, N)     x 256  inter macroblocks with one motion vector, one wavefront each
 *     k_recon_inter<1>  grid (n_quad/4, N)    x 256  inter macroblocks with one motion vector per 8x8 quadrant
 *     k_recon_inter<2>  grid (n_rest/4, N)    x 256  finer partitions
 *     k_dbk             grid (32, N)          x 256  boundary strengths from metadata; on a third HIP stream, next to the four above, joined before k_frame_dbk
 *     k_frame_intra     grid (N)              x 768  one workgroup per picture: intra macroblocks, dataflow-scheduled in LDS
 *     k_frame_dbk       grid (N)              x 768  one workgroup per picture: in-loop filter, dataflow-scheduled in LDS
 * Occupancy of the two per-picture kernels comes from batching streams: 256 pictures = one workgroup per CU.
 * Which pictures share a tick, and on which HIP stream a tick runs, is the lane scheduler's business (Lane, flush_locked;
 * for the replay sets h264bsdmiReplayCreateSched).
 *
 * This replaces, for the pixels, what the reference does synchronously inside h264bsdDecode
 * (src/h264bsd_slice_data.c:185 -> h264bsdDecodeMacroblock, src/h264bsd_decoder.c:475 ->
 * h264bsdFilterPicture) and the malloc'ed frame buffers of src/h264bsd_dpb.c:1014-1031.
 */
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <deque>
#include <mutex>
#include <vector>
#include <algorithm>
#include "kernels.hip.h"
#include "engine.h"
#include "../../include/h264bsd_mi355x_bench.h"

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            fprintf(stderr, "h264bsd-mi355x: HIP error %s at %s:%d (%s)\n", hipGetErrorString(e_), \
                    __FILE__, __LINE__, #expr);                                                    \
            return -1;                                                                             \
        }                                                                                          \
    } while (0)

namespace {

constexpr int MAX_DEVICES = 16;
struct PendingJob { uint8_t *host; uint32_t bytes, cap; const uint8_t *dev; };     /* dev: the device's address of the pinned buffer (k_h2d) */
constexpr size_t MAX_QUEUED_PICTURES = 8;    /* per decoder instance, before sink_submit starts the device on its own */

struct StreamCtx {
    uint32_t wmb = 0, hmb = 0, n_slots = 0, frame_bytes = 0;
    uint8_t *d_frames = nullptr;
    uint8_t *d_dbk = nullptr;
    uint8_t *h_frame[FJ_MAX_SLOTS] = {};      /* pinned host mirrors of the frame buffers, written by the layout kernel itself ... */
    uint8_t *hd_frame[FJ_MAX_SLOTS] = {};     /* ... through these device pointers (no staging copy in HBM, no copy engine: out_begin) */
    int out_slot = -1;                      /* the frame buffer whose picture is on its way to host memory (sink_fetch_begin .. sink_fetch_end) */
    std::vector<uint8_t *> retired_host;    /* host mirrors of the frame buffers a new parameter set replaced: a picture pulled just before lives in one (freed at the next pull) */
    uint32_t *h_conv = nullptr, *hd_conv = nullptr, *d_conv = nullptr;      /* converted picture: host mirror + its device pointer; device copy (device-resident output) */
    hipEvent_t out_ev = nullptr;            /* recorded behind the copy of a picture on its way out: the caller waits for it OUTSIDE the engine's mutex */
    std::mutex qmu;                         /* guards pending / free_bufs (submit runs on the caller's threads) */
    PendingJob acquired = { nullptr, 0, 0, nullptr };   /* staging buffer the parser is currently filling (sink_acquire) */
    std::deque<PendingJob> pending;
    std::vector<PendingJob> free_bufs;      /* recycled pinned staging buffers */
    /* lane scheduling (flush_locked): the light lane this instance belongs to, where its latest picture was launched
     * (lane index, launch number on that lane) and the first round of the current flush that may take its next one */
    int group = 0, last_lane = -1;
    unsigned long long last_launch = 0;
    unsigned ready_round = 0;
    size_t flush_quota = 0;                 /* jobs that were queued when the current flush began: it takes no others (flush_locked) */
};

/* k_dbk (boundary strengths) needs only the frame job, not pixels: it runs on a second HIP stream next to the
 * reconstruction kernels of the same tick and joins before k_frame_dbk (measured: 245.2 -> 237.5 ms per step).  k_copy runs on a
 * THIRD stream beside both and joins before k_frame_intra (round 6: 87.5-88.5 -> 84.4-85.0 ms per step; on the SAME stream as
 * k_dbk, in front of it or behind it, it gains a third of that; k_recon_inter<1> behind it on that stream loses 2.4 ms again).  Other placements that were measured — k_dbk forked behind k_copy or
 * behind the inter kernels, k_dbk of the NEXT tick next to this tick's per-picture kernels — kept the sum or lost
 * (docs/EXPERIMENTS.md); those variants are not in the product. */
struct SideLane {
    hipStream_t stream = nullptr; hipEvent_t fork = nullptr, join = nullptr;            /* k_dbk */
    hipStream_t copy_stream = nullptr; hipEvent_t copy_join = nullptr;                 /* k_copy (launch_tick) */
    bool create(int dbk_priority, bool with_priority)
    {
        if ((with_priority ? hipStreamCreateWithPriority(&stream, hipStreamNonBlocking, dbk_priority) : hipStreamCreateWithFlags(&stream, hipStreamNonBlocking)) != hipSuccess) return false;
        return hipEventCreateWithFlags(&fork, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&join, hipEventDisableTiming) == hipSuccess &&
               hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&copy_join, hipEventDisableTiming) == hipSuccess;
    }
    void destroy()
    {
        if (stream) hipStreamDestroy(stream);
        if (copy_stream) hipStreamDestroy(copy_stream);
        if (fork) hipEventDestroy(fork);
        if (join) hipEventDestroy(join);
        if (copy_join) hipEventDestroy(copy_join);
        *this = SideLane();
    }
};

/* A lane = one HIP stream that runs ticks one after the other, with its own device arena for the frame jobs of a tick
 * and its own descriptor staging.  Light lanes: one per stream group (decoder instances are dealt round-robin to the
 * groups), so that the per-picture kernels of one group — which last as long as the slowest picture of the tick —
 * overlap with the work of the other groups.  Heavy lanes: pictures that are mostly intra coded take 3-4 times as long
 * as the others in the per-picture kernels; they leave their group's tick and run on one of these, and the instance
 * rejoins its group HEAVY_DELAY rounds later if it has that many pictures queued.  Order inside one instance is kept by
 * events: every launch records ring[launch number % RING] on its lane, and a picture whose predecessor ran on another
 * lane makes its lane wait for that event (a recycled ring slot stands for a later launch of the same lane, which only
 * waits longer).  Measured on the replay sets: 256 desynchronised 1080p streams, 304 M MB/s with common
 * ticks, 550 M with heavy lanes, 660 M with 8 groups + 4 heavy lanes; more than ~12 busy HIP streams fall off a cliff
 * on this runtime (tools/probes/queue_probe.hip), and the runtime's default of 4 hardware queues serialises lanes:
 * the library asks for 16 (GPU_MAX_HW_QUEUES) when it is loaded before the HIP runtime starts. */
struct Lane {
    hipStream_t st = nullptr;
    bool owns_stream = false;
    uint8_t *d_arena = nullptr; size_t arena_cap = 0;      /* device copies of the blobs of one tick */
    FrameDesc *d_desc = nullptr, *h_desc = nullptr; size_t desc_cap = 0;    /* h_desc: pinned staging, 2 halves */
    h264k::H2dItem *h_items = nullptr, *dv_items = nullptr;                    /* pinned, 2 halves like h_desc: the tick's jobs for k_h2d (dv_: the device's view of it) */
    int flip = 0; unsigned ticks = 0;
    hipEvent_t desc_ev[2] = { nullptr, nullptr };
    static constexpr unsigned RING = 64;
    hipEvent_t ring[RING] = {};
    unsigned long long launches = 0;
    hipEvent_t tail = nullptr;
    const SideLane *side = nullptr;
};
constexpr unsigned HEAVY_DELAY = 4;
constexpr unsigned MAX_LANES_TOTAL = 10;     /* light + heavy lanes of an engine (the engine's own stream and the side stream come on top) */
#ifndef CONVERT_WGS_N
#define CONVERT_WGS_N 128       /* a wavefront converts batches of CONV_BATCH tile pairs: 128 workgroups x 4 wavefronts = the 510 batches of a 1080p picture */
#endif
constexpr unsigned CONVERT_WGS = CONVERT_WGS_N;      /* workgroups per picture of k_convert_tiles in batched launches */
#ifndef CONV_WAVES_N
#define CONV_WAVES_N 4          /* wavefronts of a k_frame_dbk workgroup that convert another picture before they filter (TickShape.conv) */
#endif
#define CONVERT_GRID(n) dim3(CONVERT_WGS, (n))
/* Under lane scheduling a k_frame_dbk workgroup shares its compute unit with the other lanes' kernels: 8 wavefronts
 * hold less of the register file than the 12 that are best when a tick has the GPU to itself (desynchronised replay
 * with 9 groups: 763 vs 734 M MB/s; lock-step, single lane: 12 wavefronts 54.9 ms per step, 8: 56.8). */
#ifndef LANE_DBK_WAVES_N
#define LANE_DBK_WAVES_N 8
#endif
constexpr uint32_t LANE_DBK_WAVES = LANE_DBK_WAVES_N;

struct Engine {
    std::mutex mu;
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t out_stream = nullptr;        /* pictures on their way to host memory (out_begin): behind the picture's OWN tick, not behind every lane */
    std::vector<Lane> lanes;                 /* [0, n_light) light lanes, then n_heavy heavy lanes */
    unsigned n_light = 1, n_heavy = 0, heavy_rr = 0, group_rr = 0;
    std::vector<StreamCtx *> streams;
    uint8_t *conv_in = nullptr; uint32_t *conv_out = nullptr; size_t conv_cap = 0; /* eng_convert_host scratch */
    std::vector<std::pair<StreamCtx *, PendingJob>> inflight;   /* staging buffers of enqueued, unfinished ticks */
    hipEvent_t inflight_done = nullptr;
    bool inflight_recorded = false;          /* inflight_done has been recorded behind everything in `inflight` */
    SideLane side;
    /* device error word (DEVERR_* bits, kernels.hip.h): the kernels OR into d_err, poll_errors() folds it into `errors`
     * whenever the host has waited for the device anyway */
    uint32_t *d_err = nullptr, *h_err = nullptr, *hd_err = nullptr;      /* device error words, their pinned host copy, the device pointer of that copy */
    uint32_t errors = 0;                       /* sticky bits (written with __atomic_fetch_or under mu, read without the lock) */
    uint32_t error_events = 0;                 /* how often a tripwire fired, ever: monotonic, so that a NEW occurrence of a bit that is
                                                  already set is visible (per-decoder copy-elision guard, the tests' delta) */
    unsigned long long *tail_prof = nullptr;   /* debug: per-wave cycle accounting of the per-picture kernels (block 0) */
};

/* One engine per HIP device, created on first use.  A decoder instance (or replay set) lives on the device that is
 * current for the thread that initialises it: h264bsdmiSetDevice() sets it for the calling thread (and, the first time,
 * the process default); one process can therefore drive several GPUs, and the usual one-process-per-GPU launch just
 * calls it once. */
Engine *g_engines[MAX_DEVICES] = {};
std::mutex g_engine_mu;
int g_default_device = -1;
thread_local int tl_device = -1;

int current_device()
{
    if (tl_device >= 0) return tl_device;
    if (g_default_device >= 0) return g_default_device;
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) d = 0;
    return d;
}

/* The HIP runtime maps streams onto 4 hardware queues unless told otherwise, and it reads the setting when it starts
 * (first HIP call of the process): lanes that share a queue run one after the other.  Ask for 16 unless the
 * application has chosen a value itself.  When the application has started the runtime before this library is loaded
 * this comes too late; set GPU_MAX_HW_QUEUES=16 in the environment then. */
__attribute__((constructor)) static void lib_init() { setenv("GPU_MAX_HW_QUEUES", "16", 0); }

/* Lane configuration: H264BSDMI_LANES="<groups>,<heavy lanes>".  Default: 4,2 when HIP streams are found to run side
 * by side (streams_run_concurrently), else one lane = the engine's own stream. */
/* Do HIP streams really run side by side here?  The answer depends on how many hardware queues the runtime was started
 * with, which the library cannot see (and can only influence when it is loaded first): it is measured.  One workgroup
 * that spins for ~500 us on each of eight fresh streams: side by side they take about as long as one, on shared queues
 * several times as long.  Returns 1 when the eight ran concurrently. */
/* the device's two error words into their pinned host copy, in stream order behind a picture on its way out (out_end_locked) */
__global__ void k_err_words(const uint32_t *__restrict__ d_err, uint32_t *__restrict__ host) { if (threadIdx.x < 2) host[threadIdx.x] = d_err[threadIdx.x]; }
__global__ void k_spin(long long ticks) { const long long t0 = wall_clock64(); while (wall_clock64() - t0 < ticks) { } }

static int streams_run_concurrently(Engine *e)
{
    constexpr int N = 8;
    hipStream_t st[N] = {};
    hipEvent_t ev[2] = {};
    int ok = 1;
    float one = 0, all = 0;
    for (auto &s : st) if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) ok = 0;
    for (auto &v : ev) if (hipEventCreate(&v) != hipSuccess) ok = 0;
    int rate_khz = 100000;                                         /* wall_clock64 ticks at 100 MHz on gfx9 */
    if (hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, e->device) != hipSuccess || rate_khz <= 0) rate_khz = 100000;
    const long long ticks = (long long)rate_khz * 500 / 1000;      /* 500 us: long against the ~0.15 ms of event traffic around the eight */
    if (ok) {
        /* warm-up: code object load, and the hardware queue behind every stream (created on first use) */
        for (auto &s : st) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, 1);
        for (auto &s : st) ok = ok && hipStreamSynchronize(s) == hipSuccess;
    }
    if (ok) {
        hipEventRecord(ev[0], st[0]);
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st[0], ticks);
        hipEventRecord(ev[1], st[0]);
        ok = hipEventSynchronize(ev[1]) == hipSuccess && hipEventElapsedTime(&one, ev[0], ev[1]) == hipSuccess;
    }
    if (ok) {
        /* st[0] starts the clock, waits for the other five, stops it */
        hipEvent_t done[N] = {};
        hipEventRecord(ev[0], st[0]);
        for (int i = 1; i < N; i++) hipStreamWaitEvent(st[i], ev[0], 0);
        for (int i = 0; i < N; i++) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st[i], ticks);
        for (int i = 1; i < N; i++) {
            if (hipEventCreateWithFlags(&done[i], hipEventDisableTiming) != hipSuccess) { ok = 0; break; }
            hipEventRecord(done[i], st[i]);
            hipStreamWaitEvent(st[0], done[i], 0);
        }
        hipEventRecord(ev[1], st[0]);
        ok = ok && hipEventSynchronize(ev[1]) == hipSuccess && hipEventElapsedTime(&all, ev[0], ev[1]) == hipSuccess;
        for (auto &d : done) if (d) hipEventDestroy(d);
    }
    for (auto &s : st) if (s) hipStreamDestroy(s);     /* (probe streams never launch banded kernels: no ticket counters) */
    for (auto &v : ev) if (v) hipEventDestroy(v);
    if (getenv("H264BSDMI_TRACE_LANES")) fprintf(stderr, "h264bsd-mi355x: stream probe: one %.3f ms, eight %.3f ms, ok %d\n", one, all, ok);
    return ok && one > 0 && all < 1.7f * one;       /* measured: 16 hardware queues 1.24 x, 4 queues 3.1 x */
}

static int lanes_create(Engine *e)
{
    unsigned g = 1, k = 0;
    if (!getenv("H264BSDMI_LANES")) {
        if (streams_run_concurrently(e)) { g = 4; k = 2; }
        else fprintf(stderr, "h264bsd-mi355x: HIP streams do not run side by side in this process (the runtime was probably started with its "
                             "default of 4 hardware queues before this library was loaded): one lane; set GPU_MAX_HW_QUEUES=16\n");
    }
    if (const char *cfg = getenv("H264BSDMI_LANES")) {
        unsigned a = 0, b = 0;
        if (sscanf(cfg, "%u,%u", &a, &b) >= 1 && a >= 1) {
            /* whatever is asked for, stay below the cliff: more than ~12 busy HIP streams (lanes + the engine's own two) make
             * this runtime crawl (19-53 s instead of 0.2 s per lap, tools/probes/queue_probe.hip) */
            g = std::min(a, 8u); k = std::min(b, 4u);
            while (g + k > MAX_LANES_TOTAL && g > 1) g--;
            if (g != a || k != b) fprintf(stderr, "h264bsd-mi355x: H264BSDMI_LANES=%s clamped to %u,%u (at most %u lanes)\n", cfg, g, k, MAX_LANES_TOTAL);
        } else fprintf(stderr, "h264bsd-mi355x: H264BSDMI_LANES=%s ignored (expected <groups>,<heavy lanes>)\n", cfg);
    }
    std::vector<Lane> lanes(g + k);                        /* (handed to the engine only when complete) */
    for (unsigned i = 0; i < g + k; i++) {
        Lane &l = lanes[i];
        if (g == 1 && i == 0) { l.st = e->stream; l.side = &e->side; }      /* the single-lane engine: k_dbk next to the reconstruction kernels */
        else { HIP_TRY(hipStreamCreateWithFlags(&l.st, hipStreamNonBlocking)); l.owns_stream = true; }
        HIP_TRY(hipEventCreateWithFlags(&l.desc_ev[0], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&l.desc_ev[1], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&l.tail, hipEventDisableTiming));
        for (auto &ev : l.ring) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    }
    e->n_light = g; e->n_heavy = k;
    e->lanes.swap(lanes);
    return 0;
}

Engine *engine_get(int device = -1)
{
    if (device < 0) device = current_device();
    if (device >= MAX_DEVICES) return nullptr;
    std::lock_guard<std::mutex> lk(g_engine_mu);
    if (g_engines[device]) return g_engines[device];
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device >= n) return nullptr;
    Engine *e = new Engine();
    e->device = device;
    if (hipSetDevice(e->device) != hipSuccess) { delete e; return nullptr; }
    if (hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess) { delete e; return nullptr; }
    if (hipEventCreateWithFlags(&e->inflight_done, hipEventDisableTiming) != hipSuccess ||
        ![&] {   /* the side stream (k_dbk next to the copy and inter kernels) gets the highest priority the device has: its few workgroups
                   * must not queue behind the hundred thousand of k_recon_inter */
            int lo = 0, hi = 0;
            const bool prio = hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess;
            return e->side.create(hi, prio);
        }() ||
        hipMalloc((void **)&e->d_err, 256) != hipSuccess || hipMemset(e->d_err, 0, 256) != hipSuccess || hipDeviceSynchronize() != hipSuccess ||   /* (hipMemset returns before the fill has run) */
        hipHostMalloc((void **)&e->h_err, 64, hipHostMallocDefault) != hipSuccess) { delete e; return nullptr; }
    e->h_err[0] = e->h_err[1] = 0;
    if (hipHostGetDevicePointer((void **)&e->hd_err, e->h_err, 0) != hipSuccess) { delete e; return nullptr; }
    g_engines[device] = e;
    return e;
}

/* ---- how the two per-picture kernels split a picture (row bands, kernels.hip.h) ----
 * rows per band for light / heavy pictures (heavy = more than a quarter of the macroblocks intra coded), wavefronts per
 * workgroup, and a cap on the bands of one launch.  H264BSDMI_TAIL="dbk_rows_light,dbk_rows_heavy,dbk_waves,intra_rows_light,
 * intra_rows_heavy,intra_waves" overrides the defaults (0 rows = one band); h264bsdmiDebugSetTail() does the same for tests. */
struct TailConfig {
    uint32_t dbk_rows_light = 17, dbk_rows_heavy = 9, dbk_waves = 12;
    uint32_t dbk_chroma_waves = 0;        /* wavefronts of a k_frame_dbk workgroup that start on the chroma graph; 0 = five twelfths */
    uint32_t intra_rows_light = 0, intra_rows_heavy = 9, intra_waves = 12;
    /* A picture is only split where that puts idle compute units to work: a launch gets at most band_budget workgroups
     * (bands per picture <= band_budget / pictures of the tick, at least 1).  256 pictures in lock-step: one workgroup per
     * picture and compute unit (measured: 4 bands x 4 wavefronts 92 instead of 57 ms per step in k_frame_dbk — a picture's
     * filtering needs about one compute unit's worth of instruction issue whichever way it is cut); the small ticks of
     * stream groups and heavy lanes: several workgroups per picture.  H264BSDMI_BAND_BUDGET overrides. */
    uint32_t band_budget = 320;
    /* ... and the heavy pictures of a tick that has the device to itself share heavy_budget further workgroups: the compute
     * units the tick's light pictures leave idle long before its heavy ones are done (H264BSDMI_HEAVY_BUDGET) */
    uint32_t heavy_budget = 64;
    bool from_env = false;
};
TailConfig g_tail;
std::mutex g_tail_mu;
TailConfig tail_config()
{
    std::lock_guard<std::mutex> lk(g_tail_mu);
    if (!g_tail.from_env) {
        g_tail.from_env = true;
        if (const char *cfg = getenv("H264BSDMI_TAIL")) {
            unsigned v[7] = { 0, 0, 0, 0, 0, 0, 0 };
            if (sscanf(cfg, "%u,%u,%u,%u,%u,%u,%u", &v[0], &v[1], &v[2], &v[3], &v[4], &v[5], &v[6]) >= 6) {
                g_tail.dbk_rows_light = v[0]; g_tail.dbk_rows_heavy = v[1]; g_tail.dbk_waves = v[2];
                g_tail.intra_rows_light = v[3]; g_tail.intra_rows_heavy = v[4]; g_tail.intra_waves = v[5];
                g_tail.dbk_chroma_waves = v[6];          /* optional seventh number; 0 = five twelfths of dbk_waves */
            } else fprintf(stderr, "h264bsd-mi355x: H264BSDMI_TAIL=%s ignored (expected six or seven numbers)\n", cfg);
        }
        if (const char *cfg = getenv("H264BSDMI_BAND_BUDGET")) g_tail.band_budget = (uint32_t)strtoul(cfg, nullptr, 10);
        if (const char *cfg = getenv("H264BSDMI_HEAVY_BUDGET")) g_tail.heavy_budget = (uint32_t)strtoul(cfg, nullptr, 10);
    }
    return g_tail;
}
static uint16_t bands_for(uint32_t hmb, uint32_t rows)
{
    if (!rows || rows >= hmb) return 1;
    return (uint16_t)std::min<uint32_t>((hmb + rows - 1) / rows, 64u);
}

/* ticket counters of the per-picture kernels (take_ticket, kernels.hip.h): one set per HIP stream that launches ticks —
 * launches on one stream run one after the other, launches on different streams may overlap.  [0,1] k_frame_dbk,
 * [2,3] k_frame_intra. */
std::mutex g_ticket_mu;
std::vector<std::pair<hipStream_t, uint32_t *>> g_tickets[MAX_DEVICES];
static uint32_t *tickets_for(hipStream_t st)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return nullptr;
    std::lock_guard<std::mutex> lk(g_ticket_mu);
    for (auto &p : g_tickets[dev]) if (p.first == st) return p.second;
    /* zeroed ON the stream whose kernels count in it: hipMemset() on device memory returns before the fill has run, on the
     * null stream, which the (non-blocking) lane streams do not wait for — a fill that landed in the middle of the first
     * banded launch handed out tickets twice and left a band of a picture without a workgroup (rare, under load only) */
    uint32_t *d = nullptr;
    if (hipMalloc((void **)&d, 64) != hipSuccess || hipMemsetAsync(d, 0, 64, st) != hipSuccess) return nullptr;
    g_tickets[dev].emplace_back(st, d);
    return d;
}
/* a stream is about to be destroyed: its counters go with it (a later stream may get the same handle and must not inherit them) */
static void tickets_release(hipStream_t st)
{
    std::lock_guard<std::mutex> lk(g_ticket_mu);
    for (auto &v : g_tickets)
        for (size_t i = 0; i < v.size();)
            if (v[i].first == st) { hipFree(v[i].second); v.erase(v.begin() + (long)i); } else i++;
}
/* a banded launch that ended early (a scheduler gave up: DEVERR_*_SCHED) leaves its ticket counters non-zero, and every later
 * banded launch on that stream would map tickets to the wrong picture and band: after a device error all of them start over */
static void tickets_rezero(int dev)
{
    std::lock_guard<std::mutex> lk(g_ticket_mu);
    if (dev < 0 || dev >= MAX_DEVICES) return;
    for (auto &p : g_tickets[dev]) (void)hipMemsetAsync(p.second, 0, 64, p.first);
}

/* ---- launch of one tick ---- */
struct TickShape {
    uint32_t n_frames = 0, max_mbs = 0;
    uint32_t max_copy = 0, max_gen = 0, max_gen_uni = 0, max_gen_quad = 0, max_gen_rest = 0, max_dbk = 0, max_levels = 0, max_w = 0, max_h = 0;
    bool any_tail = false, any_deblock = false;
    uint32_t dbk_waves = 0;          /* wavefronts per workgroup of k_frame_dbk; 0 = the configured default (launch_tick) */
    /* row bands of the two per-picture kernels: most bands a light / a heavy picture of the tick wants, for k_frame_dbk [0]
     * and k_frame_intra [1]; number of heavy pictures (more than a quarter of the macroblocks intra coded) */
    uint32_t want_light[2] = { 1, 1 }, want_heavy[2] = { 1, 1 }, n_heavy = 0;
    /* a picture of the tick whose intra schedule may wait for macroblocks BELOW (concealment, FjHeader.intra_down_deps) must
     * stay in ONE band of k_frame_intra: the launch's rows-per-band cap (which the kernel applies to every picture) must
     * then cover a whole picture, whatever the other pictures of the tick want */
    bool intra_whole = false;
    uint32_t load = 0;               /* pictures the device works on at the same time as this tick (other lanes' ticks included): the
                                        band budget is shared between them; 0 = this tick only */
    /* hosted colour conversion (FrameDesc.conv_*): descriptors of this tick name a finished picture to convert */
    bool conv = false;
    uint32_t conv_waves = 0;         /* wavefronts of a k_frame_dbk workgroup that convert before they filter; 0 = CONV_WAVES_N */
};

/* descriptor of one picture: device addresses of the sections of its (device-resident) frame job */
void make_desc(FrameDesc &d, const uint8_t *host_blob, const uint8_t *dev_blob, uint8_t *dev_frames, uint32_t frame_bytes,
               uint8_t *dev_dbk, TickShape *shape, uint32_t *dev_err)
{
    const FjHeader *h = reinterpret_cast<const FjHeader *>(host_blob);
    d.recs = reinterpret_cast<const FjMbRec *>(dev_blob + h->rec_off);
    d.mvx = reinterpret_cast<const int16_t *>(dev_blob + h->mvx_off);
    d.coefs = reinterpret_cast<const int16_t *>(dev_blob + h->coef_off);
    d.lvl = reinterpret_cast<const uint32_t *>(dev_blob + h->lvl_off);
    d.idx = reinterpret_cast<const uint16_t *>(dev_blob + h->idx_off);
    d.copy = reinterpret_cast<const FjCopy *>(dev_blob + h->copy_off);
    d.gen = reinterpret_cast<const FjGen *>(dev_blob + h->gen_off);
    d.n_copy = h->n_copy;
    d.n_gen = h->n_gen;
    d.n_gen_uni = h->n_gen_uniform;
    d.n_gen_quad = h->n_gen_quad;
    d.dbki = reinterpret_cast<const uint16_t *>(dev_blob + h->dbk_off);
    d.n_dbk = h->n_dbk;
    d.dbk = dev_dbk;
    d.cur = dev_frames + (size_t)h->cur_slot * frame_bytes;
    d.n_mbs = h->n_mbs;
    d.n_levels = h->n_intra_levels;
    d.wmb = h->width_mbs;
    d.wmb_magic = (uint32_t)(0x100000000ull / (h->width_mbs ? h->width_mbs : 1u)) + 1u;
    d.hmb = h->height_mbs;
    d.any_deblock = h->any_deblock;
    {
        const TailConfig tc = tail_config();
        const bool heavy = h->n_intra * 4u > h->n_mbs;
        d.heavy = heavy ? 1 : 0;
        d.dbk_bands = bands_for(h->height_mbs, heavy ? tc.dbk_rows_heavy : tc.dbk_rows_light);
        /* a concealed macroblock may wait for the macroblocks below it (FJ_NEED_D*): such a picture stays in one band,
         * a band never waits for a band that started after it */
        d.intra_bands = h->intra_down_deps ? 1 : bands_for(h->height_mbs, heavy ? tc.intra_rows_heavy : tc.intra_rows_light);
    }
    d.err = dev_err;
    d.conv_src = nullptr; d.conv_dst = nullptr; d.conv_fmt = 0; d.conv_pad = 0;
    for (uint32_t k = 0; k < FJ_MAX_SLOTS; k++) d.slot[k] = k < h->n_slots ? dev_frames + (size_t)k * frame_bytes : nullptr;
    if (shape) {
        shape->n_frames++;
        shape->max_mbs = std::max(shape->max_mbs, h->n_mbs);
        shape->max_copy = std::max(shape->max_copy, h->n_copy);
        shape->max_gen = std::max(shape->max_gen, h->n_gen);
        shape->max_gen_uni = std::max(shape->max_gen_uni, h->n_gen_uniform);
        shape->max_gen_quad = std::max(shape->max_gen_quad, h->n_gen_quad);
        shape->max_gen_rest = std::max(shape->max_gen_rest, h->n_gen - h->n_gen_uniform - h->n_gen_quad);
        shape->max_dbk = std::max(shape->max_dbk, h->n_dbk);
        shape->any_deblock |= h->any_deblock != 0;
        shape->max_levels = std::max(shape->max_levels, h->n_intra_levels);
        shape->max_w = std::max<uint32_t>(shape->max_w, h->width_mbs);
        shape->max_h = std::max<uint32_t>(shape->max_h, h->height_mbs);
        shape->any_tail |= h->n_intra_levels != 0 || h->any_deblock != 0;
        shape->n_heavy += d.heavy;
        if (h->any_deblock) (d.heavy ? shape->want_heavy : shape->want_light)[0] = std::max<uint32_t>((d.heavy ? shape->want_heavy : shape->want_light)[0], d.dbk_bands);
        if (h->n_intra_levels) (d.heavy ? shape->want_heavy : shape->want_light)[1] = std::max<uint32_t>((d.heavy ? shape->want_heavy : shape->want_light)[1], d.intra_bands);
        if (h->n_intra_levels && h->intra_down_deps) shape->intra_whole = true;
    }
}

struct TickTimers { hipEvent_t ev[6]; hipEvent_t sev[5] = { nullptr, nullptr, nullptr, nullptr, nullptr }; bool on = false; bool copy_timed = false; unsigned mask = 31u; };   /* copy_timed: sev[3..4] were recorded by the latest launch_tick */   /* mask bit k: kernel k of KERNELS is timed */   /* boundaries of the 5 kernels of a tick; sev[1..2] = k_dbk on the side stream, sev[3..4] = k_copy on the copy stream */

/* k_dbk of one tick on the side stream (which must already wait for whatever frees the tick's deblocking scratch);
 * records the join event behind it */
static int launch_kdbk_aside(const SideLane *side, const FrameDesc *d_desc, const TickShape &s, TickTimers *tt, uint32_t launches[5], unsigned stages)
{
    const bool do_dbk = (stages & 4u) && s.any_deblock && s.max_dbk;
    const bool timed = tt && tt->on && (tt->mask & 4u);
    if (timed && tt->sev[1]) HIP_TRY(hipEventRecord(tt->sev[1], side->stream));
    if (do_dbk) {
        hipLaunchKernelGGL(h264k::k_dbk, dim3(std::min<uint32_t>((s.max_dbk + 4 * DBK_WG_WAVES - 1) / (4 * DBK_WG_WAVES), DBK_WGS * 4 / DBK_WG_WAVES), s.n_frames), dim3(64 * DBK_WG_WAVES), 0, side->stream, d_desc);
        if (launches) launches[2]++;
    }
    if (timed && tt->sev[2]) HIP_TRY(hipEventRecord(tt->sev[2], side->stream));
    HIP_TRY(hipEventRecord(side->join, side->stream));
    return 0;
}

int launch_tick(hipStream_t st, const FrameDesc *d_desc, const TickShape &s, TickTimers *tt, uint32_t launches[5],
                unsigned stages = 7u, const SideLane *side = nullptr, unsigned long long *prof = nullptr)
{
    const bool timed = tt && tt->on;
    const unsigned tmask = timed ? tt->mask : 0u;
#define EV_NEEDED(i) ((((tmask << 1) | tmask) >> (i)) & 1u)      /* boundary i closes kernel i-1 and opens kernel i */
    if (EV_NEEDED(0)) HIP_TRY(hipEventRecord(tt->ev[0], st));
    const bool do_dbk = (stages & 4u) && s.any_deblock && s.max_dbk;
    const bool do_copy = (stages & 1u) && s.max_copy;
    const bool aside = side && side->stream && do_dbk;
    /* k_copy on a stream of its own beside k_recon_inter (and k_dbk): the copies are HBM-bound, the inter kernel is bound by
     * instruction issue, and they write different macroblocks of the picture; both are joined in front of k_frame_intra, whose
     * macroblocks read them.  84.5-85.2 against 87.5-88.5 ms per step (docs/EXPERIMENTS.md, "k_copy beside the inter kernels"). */
    const bool copy_aside = aside && do_copy && side->copy_stream;
    if (tt) tt->copy_timed = false;
    auto launch_copy = [&](hipStream_t cs) {
        hipLaunchKernelGGL(h264k::k_copy, dim3(std::min<uint32_t>((s.max_copy + 3) / 4, COPY_WGS), s.n_frames), dim3(256), 0, cs, d_desc);   /* COPY_WGS workgroups per picture walk its run list */
        if (launches) launches[0]++;
    };
    if (aside) {
        HIP_TRY(hipEventRecord(side->fork, st));                 /* after the previous tick's k_frame_dbk: the records are free */
        HIP_TRY(hipStreamWaitEvent(side->stream, side->fork, 0));
        if (copy_aside) {
            const bool ctimed = timed && (tmask & 1u) && tt->sev[3] && tt->sev[4];
            if (tt) tt->copy_timed = ctimed;
            HIP_TRY(hipStreamWaitEvent(side->copy_stream, side->fork, 0));
            if (ctimed) HIP_TRY(hipEventRecord(tt->sev[3], side->copy_stream));
            launch_copy(side->copy_stream);
            if (ctimed) HIP_TRY(hipEventRecord(tt->sev[4], side->copy_stream));
            HIP_TRY(hipEventRecord(side->copy_join, side->copy_stream));
        }
        if (launch_kdbk_aside(side, d_desc, s, tt, launches, stages)) return -1;
    }
    if (do_copy && !copy_aside) launch_copy(st);
    if (EV_NEEDED(1)) HIP_TRY(hipEventRecord(tt->ev[1], st));
    if ((stages & 1u) && s.max_gen) {
        if (s.max_gen_uni) hipLaunchKernelGGL(h264k::k_recon_inter<0>, dim3((s.max_gen_uni + INTER_WG_WAVES * h264k::inter_per_wave<0>() - 1) / (INTER_WG_WAVES * h264k::inter_per_wave<0>()), s.n_frames), dim3(64 * INTER_WG_WAVES), 0, st, d_desc);
        if (s.max_gen_quad) hipLaunchKernelGGL(h264k::k_recon_inter<1>, dim3((s.max_gen_quad + INTER_WG_WAVES * h264k::inter_per_wave<1>() - 1) / (INTER_WG_WAVES * h264k::inter_per_wave<1>()), s.n_frames), dim3(64 * INTER_WG_WAVES), 0, st, d_desc);
        if (s.max_gen_rest) hipLaunchKernelGGL(h264k::k_recon_inter<2>, dim3((s.max_gen_rest + INTER_WG_WAVES * h264k::inter_per_wave<2>() - 1) / (INTER_WG_WAVES * h264k::inter_per_wave<2>()), s.n_frames), dim3(64 * INTER_WG_WAVES), 0, st, d_desc);
        if (launches) launches[1]++;
    }
    if (EV_NEEDED(2)) HIP_TRY(hipEventRecord(tt->ev[2], st));
    if (do_dbk && !aside) {
        hipLaunchKernelGGL(h264k::k_dbk, dim3(std::min<uint32_t>((s.max_dbk + 4 * DBK_WG_WAVES - 1) / (4 * DBK_WG_WAVES), DBK_WGS * 4 / DBK_WG_WAVES), s.n_frames), dim3(64 * DBK_WG_WAVES), 0, st, d_desc);
        if (launches) launches[2]++;
    }
    if (EV_NEEDED(3)) HIP_TRY(hipEventRecord(tt->ev[3], st));
    /* The two per-picture kernels keep per-macroblock scheduling state in LDS next to their wavefronts' tiles: for
     * pictures that leave less than 16 wavefronts' worth of tile space in the 160 KB of a CU, fewer wavefronts run. */
    constexpr size_t LDS_BUDGET = 160 * 1024 - 512;
    /* Row bands of the per-picture kernels (kernels.hip.h).  A picture is split only where that puts IDLE compute units to
     * work — measured: with 256 pictures on 256 compute units every split loses (a picture's work is about one compute unit's
     * worth of instruction issue however it is cut: 4 bands x 4 wavefronts 92 instead of 57 ms per step in k_frame_dbk, bands on
     * the heavy lanes of a saturated desynchronised schedule 775 instead of 827 M MB/s); with few pictures on the device it wins
     * (4-32 streams: +18-20 %), and so it does for the few heavy pictures of a tick that is otherwise done long before them.
     *   light_cap: bands a light picture may use = band_budget / pictures on the device (this tick, or all lanes' ticks: load)
     *   heavy_cap: when the tick is (nearly) alone on the device, its heavy pictures share what the budget leaves */
    const TailConfig tc = tail_config();
    const uint32_t on_device = std::max<uint32_t>(1u, std::max(s.n_frames, s.load));
    const uint32_t light_cap = std::max<uint32_t>(1u, tc.band_budget / on_device);
    uint32_t heavy_cap = light_cap;
    if (s.n_heavy && 2u * s.n_frames >= s.load) heavy_cap = std::max(light_cap, 1u + tc.heavy_budget / s.n_heavy);
    struct BandPlan { uint32_t bands, rows, waves; size_t lds; };
    auto plan = [&](int which, uint32_t waves, size_t (*lds_bytes)(uint32_t, uint32_t, uint32_t), bool may_shorten, BandPlan &bp) -> int {
        const uint32_t eff_l = std::min(s.want_light[which], light_cap), eff_h = std::min(s.want_heavy[which], heavy_cap);
        /* rows a band can have: the picture with the fewest bands decides (all pictures of a tick have the tick's size in
         * practice; max_h / fewest bands is the bound) */
        uint32_t fewest = s.n_heavy >= s.n_frames ? eff_h : s.n_heavy ? std::min(eff_l, eff_h) : eff_l;
        /* (band_split() clamps a picture's rows per band to this cap: a picture that wants ONE band gets it only if the cap is
         * the picture's height — for k_frame_intra that is a matter of correctness, see TickShape::intra_whole) */
        if (which == 1 && s.intra_whole) fewest = 1;
        uint32_t rows = (s.max_h + fewest - 1) / std::max<uint32_t>(1u, fewest);
        rows = std::max<uint32_t>(1u, std::min<uint32_t>(rows, s.max_h));
        while (may_shorten && lds_bytes(waves, s.max_w, rows) > LDS_BUDGET && rows > 1) rows = (rows + 1) / 2;      /* (rows is a cap the kernel applies to every picture) */
        while (lds_bytes(waves, s.max_w, rows) > LDS_BUDGET && waves > 1) waves--;
        if (lds_bytes(waves, s.max_w, rows) > LDS_BUDGET) return -1;
        bp.rows = rows; bp.waves = waves; bp.lds = lds_bytes(waves, s.max_w, rows);
        bp.bands = std::max<uint32_t>(std::max(s.n_heavy < s.n_frames ? eff_l : 1u, s.n_heavy ? eff_h : 1u), (s.max_h + rows - 1) / rows);
        return 0;
    };
    if (copy_aside) HIP_TRY(hipStreamWaitEvent(st, side->copy_join, 0));      /* (the copies have to be there before the intra macroblocks read them) */
    if (s.max_levels && (stages & 2u)) {
        BandPlan bp;
        /* a picture with concealed macroblocks must stay in one band (FjHeader.intra_down_deps): fewer wavefronts, never
         * shorter bands */
        if (plan(1, std::max<uint32_t>(1u, std::min<uint32_t>(tc.intra_waves, h264k::TAIL_WAVES)), h264k::intra_lds_bytes, false, bp)) {
            fprintf(stderr, "h264bsd-mi355x: picture of %u x %u macroblocks is too large for k_frame_intra\n", s.max_w, s.max_h); return -1;
        }
        const uint32_t bands = bp.bands, rows = bp.rows, waves = bp.waves;
        const size_t lds = bp.lds;
        uint32_t *tickets = bands > 1 ? tickets_for(st) : nullptr;
        if (bands > 1 && !tickets) return -1;
        int dev = 0;
        HIP_TRY(hipGetDevice(&dev));
        static size_t lds_enabled[2][MAX_DEVICES] = {};
        static std::mutex lds_mu;
        {
            std::lock_guard<std::mutex> lk(lds_mu);
            if (lds > lds_enabled[bands > 1][dev]) {
                HIP_TRY(hipFuncSetAttribute(bands > 1 ? (const void *)h264k::k_frame_intra<true> : (const void *)h264k::k_frame_intra<false>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                lds_enabled[bands > 1][dev] = lds;
            }
        }
        if (bands > 1) hipLaunchKernelGGL(h264k::k_frame_intra<true>, dim3(s.n_frames * bands), dim3(64 * waves), lds, st, d_desc, prof, tickets + 2, bands, rows, light_cap);
        else hipLaunchKernelGGL(h264k::k_frame_intra<false>, dim3(s.n_frames), dim3(64 * waves), lds, st, d_desc, prof, nullptr, 1u, rows, 1u);
        if (launches) launches[3]++;
    }
    if (EV_NEEDED(4)) HIP_TRY(hipEventRecord(tt->ev[4], st));
    if (aside) HIP_TRY(hipStreamWaitEvent(st, side->join, 0));
    if (s.any_deblock && (stages & 4u)) {
        BandPlan bp;
        if (plan(0, std::max<uint32_t>(1u, std::min<uint32_t>(s.dbk_waves ? s.dbk_waves : tc.dbk_waves, h264k::DBK_WAVES)), h264k::dbk_lds_bytes, true, bp)) {
            fprintf(stderr, "h264bsd-mi355x: picture %u macroblocks wide is too large for k_frame_dbk\n", s.max_w); return -1;
        }
        const uint32_t bands = bp.bands, rows = bp.rows, waves = bp.waves;
        /* five twelfths of a workgroup's wavefronts start on the chroma graph (k_frame_dbk.hip.h); the seventh number of H264BSDMI_TAIL overrides */
        const uint32_t chroma_waves = tc.dbk_chroma_waves ? tc.dbk_chroma_waves : std::max<uint32_t>(1u, (waves * 5u + 6u) / 12u);     /* 12 -> 5 (4 / 5 / 6: 31.6 / 30.3 / 30.3 ms per step), 8 -> 3 */
        const size_t lds = bp.lds;
        uint32_t *tickets = bands > 1 ? tickets_for(st) : nullptr;
        if (bands > 1 && !tickets) return -1;
        int dev = 0;
        HIP_TRY(hipGetDevice(&dev));
        static size_t lds_enabled[2][MAX_DEVICES] = {};
        static std::mutex lds_mu;
        {
            std::lock_guard<std::mutex> lk(lds_mu);
            if (lds > lds_enabled[bands > 1][dev]) {
                HIP_TRY(hipFuncSetAttribute(bands > 1 ? (const void *)h264k::k_frame_dbk<true> : (const void *)h264k::k_frame_dbk<false>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                lds_enabled[bands > 1][dev] = lds;
            }
        }
        const uint32_t conv_waves = !s.conv ? 0u : std::min<uint32_t>(s.conv_waves ? s.conv_waves : (uint32_t)CONV_WAVES_N, waves > chroma_waves + 1u ? waves - chroma_waves - 1u : 0u);
        if (bands > 1) hipLaunchKernelGGL(h264k::k_frame_dbk<true>, dim3(s.n_frames * bands), dim3(64 * waves), lds, st, d_desc, prof, tickets, bands, rows, light_cap, chroma_waves, conv_waves);
        else hipLaunchKernelGGL(h264k::k_frame_dbk<false>, dim3(s.n_frames), dim3(64 * waves), lds, st, d_desc, prof, nullptr, 1u, rows, 1u, chroma_waves, conv_waves);
        if (launches) launches[4]++;
    }
    if (s.conv && !(s.any_deblock && (stages & 4u)))             /* no k_frame_dbk in this tick: nobody hosted the conversion */
        hipLaunchKernelGGL(h264k::k_convert_rest, dim3(CONVERT_WGS, s.n_frames), dim3(256), 0, st, d_desc);
    if (EV_NEEDED(5)) HIP_TRY(hipEventRecord(tt->ev[5], st));
#undef EV_NEEDED
    HIP_TRY(hipGetLastError());
    return 0;
}

/* Fold the device error word into the engine's sticky error bits.  Called where the host waits for the stream anyway. */
static void fold_errors(Engine *e)                    /* h_err holds a copy of the device's words that has arrived */
{
    const uint32_t fresh = e->h_err[0] & ~e->errors;
    /* monotonic: the pinned words are written by copies on two streams, an older snapshot may land last */
    if (e->h_err[1] > __atomic_load_n(&e->error_events, __ATOMIC_RELAXED)) __atomic_store_n(&e->error_events, e->h_err[1], __ATOMIC_RELAXED);
    if (fresh) {
        fprintf(stderr, "h264bsd-mi355x: DEVICE ERROR 0x%x:%s%s%s — pixels of the affected pictures are not trustworthy\n", fresh,
                (fresh & DEVERR_RESIDUAL_RANGE) ? " residual outside [-512,511] reached the kernels (host check missed it)" : "",
                (fresh & DEVERR_INTRA_SCHED) ? " k_frame_intra scheduler gave up" : "", (fresh & DEVERR_DBK_SCHED) ? " k_frame_dbk scheduler gave up" : "");
        __atomic_fetch_or(&e->errors, fresh, __ATOMIC_RELAXED);
        tickets_rezero(e->device);
    }
}
int poll_errors(Engine *e)
{
    HIP_TRY(hipMemcpyAsync(e->h_err, e->d_err, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    fold_errors(e);
    return 0;
}

/* ---- lazy execution of the queued jobs of all decoder instances ---- */
/* Recycle the pinned staging buffers of ticks whose copies have completed (all of them after wait != 0). */
int reap_locked(Engine *e, bool wait)
{
    if (e->inflight.empty()) return 0;
    if (wait) {
        /* (after a failed flush the event may be missing: wait for the lanes themselves before the staging buffers go back) */
        if (!e->inflight_recorded) for (auto &l : e->lanes) if (l.st) (void)hipStreamSynchronize(l.st);
        if (poll_errors(e)) return -1;
    } else if (!e->inflight_recorded || hipEventQuery(e->inflight_done) != hipSuccess) return 0;
    for (auto &f : e->inflight) {
        std::lock_guard<std::mutex> ql(f.first->qmu);
        f.first->free_bufs.push_back(f.second);
    }
    e->inflight.clear();
    return 0;
}

/* One tick on one lane: the front jobs of `part` (popped here) are copied to the lane's arena and launched. */
static int lane_launch(Engine *e, unsigned lane_idx, const std::vector<StreamCtx *> &part)
{
    Lane &l = e->lanes[lane_idx];
    size_t bytes = 0;
    for (StreamCtx *s : part) {
        std::lock_guard<std::mutex> ql(s->qmu);
        bytes += (s->pending.front().bytes + 255u) & ~255u;
    }
    if (bytes > l.arena_cap) {
        HIP_TRY(hipStreamSynchronize(l.st));               /* earlier ticks may still read the old arena */
        if (l.d_arena) HIP_TRY(hipFree(l.d_arena));
        l.arena_cap = bytes + bytes / 4;
        HIP_TRY(hipMalloc((void **)&l.d_arena, l.arena_cap));
    }
    if (part.size() > l.desc_cap) {
        HIP_TRY(hipStreamSynchronize(l.st));
        if (l.d_desc) HIP_TRY(hipFree(l.d_desc));
        if (l.h_desc) HIP_TRY(hipHostFree(l.h_desc));
        if (l.h_items) HIP_TRY(hipHostFree(l.h_items));
        l.desc_cap = part.size() * 2;
        HIP_TRY(hipMalloc((void **)&l.d_desc, l.desc_cap * sizeof(FrameDesc)));
        HIP_TRY(hipHostMalloc((void **)&l.h_desc, 2 * l.desc_cap * sizeof(FrameDesc), hipHostMallocDefault));
        HIP_TRY(hipHostMalloc((void **)&l.h_items, 2 * l.desc_cap * sizeof(h264k::H2dItem), hipHostMallocDefault));
        HIP_TRY(hipHostGetDevicePointer((void **)&l.dv_items, l.h_items, 0));
        l.flip = 0; l.ticks = 0;
    }
    /* descriptors are staged in pinned memory (two halves, alternating) so that the copy can be asynchronous;
     * before reusing a half, the tick that used it two ticks ago must have consumed it */
    if (l.ticks >= 2) HIP_TRY(hipEventSynchronize(l.desc_ev[l.flip]));
    FrameDesc *descs = l.h_desc + (size_t)l.flip * l.desc_cap;
    h264k::H2dItem *items = l.h_items + (size_t)l.flip * l.desc_cap;
    /* how the jobs reach the device: one k_h2d launch that reads the pinned staging buffers (256 hipMemcpyAsync calls per tick before) */
    TickShape shape;
    size_t off = 0;
    std::vector<std::pair<int, unsigned long long>> waited;
    for (size_t i = 0; i < part.size(); i++) {
        StreamCtx *s = part[i];
        PendingJob j;
        { std::lock_guard<std::mutex> ql(s->qmu); j = s->pending.front(); s->pending.pop_front(); }
        if (s->last_lane >= 0 && (unsigned)s->last_lane != lane_idx) {     /* its previous picture ran on another lane */
            const std::pair<int, unsigned long long> key(s->last_lane, s->last_launch);
            if (std::find(waited.begin(), waited.end(), key) == waited.end()) {
                HIP_TRY(hipStreamWaitEvent(l.st, e->lanes[s->last_lane].ring[s->last_launch % Lane::RING], 0));
                waited.push_back(key);
            }
        }
        items[i] = h264k::H2dItem{ j.dev, l.d_arena + off, j.bytes, 0u };
        make_desc(descs[i], j.host, l.d_arena + off, s->d_frames, s->frame_bytes, s->d_dbk, &shape, e->d_err);
        off += (j.bytes + 255u) & ~255u;
        e->inflight.emplace_back(s, j);
        e->inflight_recorded = false;
        s->last_lane = (int)lane_idx; s->last_launch = l.launches;
    }
    hipLaunchKernelGGL(h264k::k_h2d, dim3(h264k::H2D_CHUNKS, (uint32_t)part.size()), dim3(256), 0, l.st, l.dv_items + (size_t)l.flip * l.desc_cap);
    HIP_TRY(hipMemcpyAsync(l.d_desc, descs, part.size() * sizeof(FrameDesc), hipMemcpyHostToDevice, l.st));
    HIP_TRY(hipEventRecord(l.desc_ev[l.flip], l.st));
    l.flip ^= 1;
    l.ticks++;
    if (e->lanes.size() > 1) shape.dbk_waves = LANE_DBK_WAVES;
    shape.load = (uint32_t)e->streams.size();
    if (launch_tick(l.st, l.d_desc, shape, nullptr, nullptr, 7u, l.side)) return -1;
    HIP_TRY(hipEventRecord(l.ring[l.launches % Lane::RING], l.st));
    l.launches++;
    return 0;
}

/* Enqueue the pending jobs, round by round: a round takes at most one picture per instance; the light pictures of a
 * group form one tick on the group's lane, the heavy pictures of all groups one tick on a heavy lane (Lane, above).
 * wait: block until the pixels exist; otherwise return once everything is enqueued (the staging buffers stay owned by
 * `inflight`). */
int flush_locked(Engine *e, bool wait = true)
{
    HIP_TRY(hipSetDevice(e->device));
    if (reap_locked(e, false)) return -1;
    if (e->lanes.empty() && lanes_create(e)) return -1;      /* on first use: a process that only runs replay sets keeps its HIP streams for those */
    const bool multi = e->lanes.size() > 1;
    bool started = false;
    std::vector<std::vector<StreamCtx *>> part(e->n_light);
    std::vector<StreamCtx *> heavy;
    /* A flush enqueues what was submitted BEFORE it began.  An application may call h264bsdmiFlushAsync() on one thread while
     * its other threads are already parsing the next round (the call takes as long as enqueueing 256 jobs does): pictures that
     * arrive meanwhile wait for the next flush instead of forming straggler ticks of a few streams each. */
    for (StreamCtx *s : e->streams) { std::lock_guard<std::mutex> ql(s->qmu); s->flush_quota = s->pending.size(); }
    for (unsigned round = 0;; round++) {
        bool any_pending = false;
        for (auto &p : part) p.clear();
        heavy.clear();
        for (StreamCtx *s : e->streams) {
            std::lock_guard<std::mutex> ql(s->qmu);
            if (s->pending.empty() || !s->flush_quota) continue;
            any_pending = true;
            if (s->ready_round > round) continue;
            s->flush_quota--;
            const FjHeader *h = reinterpret_cast<const FjHeader *>(s->pending.front().host);
            if (e->n_heavy && h->n_intra * 4u > h->n_mbs) heavy.push_back(s);
            else part[(unsigned)s->group % e->n_light].push_back(s);
        }
        if (!any_pending) break;
        started = true;
        for (unsigned g = 0; g < e->n_light; g++)
            if (!part[g].empty() && lane_launch(e, g, part[g])) return -1;
        if (!heavy.empty()) {
            if (lane_launch(e, e->n_light + e->heavy_rr++ % e->n_heavy, heavy)) return -1;
            for (StreamCtx *s : heavy) s->ready_round = round + 1 + HEAVY_DELAY;
        }
    }
    for (StreamCtx *s : e->streams) s->ready_round = 0;          /* rounds count from the start of a flush */
    /* Lanes never wait for the engine's own stream: everything that stream does to frame buffers (clearing them in
     * sink_configure, laying pictures out for the application) is complete when the call that enqueued it returns. */
    if (started && multi)
        for (auto &l : e->lanes) {                               /* the engine's stream continues behind all lanes */
            HIP_TRY(hipEventRecord(l.tail, l.st));
            HIP_TRY(hipStreamWaitEvent(e->stream, l.tail, 0));
        }
    if (!e->inflight.empty()) { HIP_TRY(hipEventRecord(e->inflight_done, e->stream)); e->inflight_recorded = true; }
    return wait ? reap_locked(e, true) : 0;
}

/* ---- JobSink implementation ---- */
struct SinkUser { Engine *e; StreamCtx *s; };

static void free_retired(StreamCtx *s)
{
    for (uint8_t *p : s->retired_host) hipHostFree(p);
    s->retired_host.clear();
}

void stream_release(StreamCtx *s, bool keep_pulled = false)
{
    for (auto &j : s->pending) hipHostFree(j.host);
    s->pending.clear();
    for (auto &j : s->free_bufs) hipHostFree(j.host);
    s->free_bufs.clear();
    if (s->acquired.host) hipHostFree(s->acquired.host);
    s->acquired = PendingJob{ nullptr, 0, 0, nullptr };
    if (s->d_frames) hipFree(s->d_frames);
    if (s->d_dbk) hipFree(s->d_dbk);
    s->d_dbk = nullptr;
    /* keep_pulled (a new sequence re-allocates the frame buffers, sink_configure): h264bsdmiPullAndDecodePictureBatch hands out a picture and
     * parses on in the same call — the picture must outlive the activation of a parameter set that the parsing may bring */
    for (auto &p : s->h_frame) if (p) { if (keep_pulled) s->retired_host.push_back(p); else hipHostFree(p); p = nullptr; }
    for (auto &p : s->hd_frame) p = nullptr;
    if (!keep_pulled) free_retired(s);
    if (s->h_conv) hipHostFree(s->h_conv);
    if (s->d_conv) hipFree(s->d_conv);
    if (s->out_ev) { hipEventDestroy(s->out_ev); s->out_ev = nullptr; }
    s->d_frames = nullptr; s->h_conv = nullptr; s->hd_conv = nullptr; s->d_conv = nullptr;
}

int sink_configure(void *user, uint32_t wmb, uint32_t hmb, uint32_t n_slots)
{
    SinkUser *u = static_cast<SinkUser *>(user);
    std::lock_guard<std::mutex> lk(u->e->mu);
    if (flush_locked(u->e)) return -1;
    HIP_TRY(hipSetDevice(u->e->device));
    if (u->e->out_stream) HIP_TRY(hipStreamSynchronize(u->e->out_stream));
    stream_release(u->s, true);
    u->s->wmb = wmb; u->s->hmb = hmb; u->s->n_slots = n_slots;
    u->s->frame_bytes = fj_frame_bytes(wmb, hmb);
    const size_t total = (size_t)n_slots * u->s->frame_bytes + 256;
    HIP_TRY(hipMalloc((void **)&u->s->d_frames, total));
    HIP_TRY(hipMemsetAsync(u->s->d_frames, 0, total, u->e->stream));
    HIP_TRY(hipMalloc((void **)&u->s->d_dbk, DBK_SCRATCH_BYTES(wmb * hmb)));
    HIP_TRY(hipMemsetAsync(u->s->d_dbk, 0, DBK_SCRATCH_BYTES(wmb * hmb), u->e->stream));
    HIP_TRY(hipStreamSynchronize(u->e->stream));           /* the lanes do not order themselves behind this stream */
    u->s->last_lane = -1;
    return 0;
}

/* staging buffer for a frame job of up to `bytes`: recycled pinned memory of this stream, else a new allocation */
static int take_buffer(SinkUser *u, uint32_t bytes, PendingJob *out)
{
    PendingJob j = { nullptr, 0, 0, nullptr };
    {
        std::lock_guard<std::mutex> ql(u->s->qmu);
        for (size_t i = 0; i < u->s->free_bufs.size(); i++)
            if (u->s->free_bufs[i].cap >= bytes) {
                j = u->s->free_bufs[i];
                u->s->free_bufs.erase(u->s->free_bufs.begin() + (long)i);
                break;
            }
    }
    if (!j.host) {
        HIP_TRY(hipSetDevice(u->e->device));
        j.cap = bytes + 65536;                              /* pinned: recycled across pictures */
        HIP_TRY(hipHostMalloc((void **)&j.host, j.cap, hipHostMallocDefault));
        HIP_TRY(hipHostGetDevicePointer((void **)&j.dev, j.host, 0));
    }
    *out = j;
    return 0;
}

/* The parser builds the next frame job directly in pinned staging memory (JobSink.acquire): nothing is copied on
 * the host between parsing and the H2D transfer. */
uint8_t *sink_acquire(void *user, uint32_t bytes)
{
    SinkUser *u = static_cast<SinkUser *>(user);
    StreamCtx *s = u->s;
    if (s->acquired.host && s->acquired.cap >= bytes) return s->acquired.host;     /* previous picture was abandoned */
    if (s->acquired.host) {
        std::lock_guard<std::mutex> ql(s->qmu);
        s->free_bufs.push_back(s->acquired);
        s->acquired = PendingJob{ nullptr, 0, 0, nullptr };
    }
    if (take_buffer(u, bytes, &s->acquired)) return nullptr;
    return s->acquired.host;
}

int sink_submit(void *user, const uint8_t *blob, uint32_t bytes)
{
    /* runs on the application's / the parser pool's threads, concurrently for different decoder instances: only the
     * stream's own queue lock is taken */
    SinkUser *u = static_cast<SinkUser *>(user);
    PendingJob j;
    if (u->s->acquired.host && blob == u->s->acquired.host) {
        j = u->s->acquired;
        u->s->acquired = PendingJob{ nullptr, 0, 0, nullptr };
    } else {
        if (take_buffer(u, bytes, &j)) return -1;
        memcpy(j.host, blob, bytes);
    }
    j.bytes = bytes;
    size_t backlog;
    {
        std::lock_guard<std::mutex> ql(u->s->qmu);
        u->s->pending.push_back(j);
        backlog = u->s->pending.size();
    }
    /* An application that decodes without pulling pictures (frame skipping, decoding ahead) must not pile up pinned
     * staging buffers (worst-case sized, ~8 MB each for 1080p): past a few queued pictures the queues are enqueued on
     * the device — asynchronously, the caller does not wait for pixels — and the buffers recycle. */
    if (backlog >= MAX_QUEUED_PICTURES) {
        std::lock_guard<std::mutex> lk(u->e->mu);
        if (flush_locked(u->e, false)) return -1;
    }
    return 0;
}

/* how often a tripwire of the kernels has fired on this device, as last folded in (poll_errors runs wherever the host waits for the
 * device): no wait here.  Monotonic: a decoder compares it with the value it saw when it was created */
uint32_t sink_errors(void *user)
{
    SinkUser *u = static_cast<SinkUser *>(user);
    return __atomic_load_n(&u->e->error_events, __ATOMIC_RELAXED);
}

/* A picture leaves the device.  Under the engine's mutex only what must be ordered: the instance's queued pictures are enqueued (not
 * awaited), the layout kernel — which writes pinned host memory itself — follows behind the picture's tick, an event of the INSTANCE is
 * recorded behind it.  The wait for that event — the pixels' whole latency: the tick's kernels, 3.1 MB over PCIe — happens outside the mutex, so that the other
 * instances of the process go on submitting, flushing and pulling meanwhile (rounds 1-4 held the mutex across a synchronous
 * copy: one picture at a time for the whole process).  Reference: the zero-copy alias of src/h264bsd_decoder.c:599-646. */
#ifndef OUT_EVENT_FLAGS
#define OUT_EVENT_FLAGS (hipEventDisableTiming | hipEventBlockingSync)    /* the waiter sleeps: under a CPU quota a spinning waiter takes the time the parser threads need */
#endif
static int out_begin(SinkUser *u, bool to_host, hipStream_t *st)          /* mutex held */
{
    /* Only an instance whose OWN pictures are still queued makes the engine enqueue (everything: ticks are formed across instances).
     * One whose pictures are all on the device already leaves the other instances' queues alone — when the pulls of one round run
     * beside the parsing of the next (h264bsdmiPullAndDecodePictureBatch), a flush per pull would cut that round's tick into as many
     * straggler ticks as there are pulls. */
    Engine *e = u->e;
    StreamCtx *s = u->s;
    bool mine;
    { std::lock_guard<std::mutex> ql(s->qmu); mine = !s->pending.empty(); }
    if (mine && flush_locked(e, false)) return -1;
    if (!mine) HIP_TRY(hipSetDevice(e->device));
    if (!s->out_ev) HIP_TRY(hipEventCreateWithFlags(&s->out_ev, OUT_EVENT_FLAGS));
    *st = e->stream;                                   /* device-resident output: on the engine's stream, which is behind every lane since the last flush */
    if (to_host) {
        /* A picture bound for host memory waits for the tick that made it (or a later one of its lane), not for whatever else the engine's
         * stream is behind: instances whose tick is done hand their pictures over while other instances' ticks still run.  Nothing the
         * device does later can touch the frame before the call returns — the instance's next job is submitted after that. */
        if (!e->out_stream) HIP_TRY(hipStreamCreateWithFlags(&e->out_stream, hipStreamNonBlocking));
        if (s->last_lane >= 0) {
            hipEvent_t made = e->lanes[s->last_lane].ring[s->last_launch % Lane::RING];
            if (hipEventQuery(made) != hipSuccess) HIP_TRY(hipStreamWaitEvent(e->out_stream, made, 0));      /* (long done, usually: no barrier packet then) */
        }
        *st = e->out_stream;
    }
    return 0;
}
static int out_end_locked(SinkUser *u, hipStream_t st)  /* mutex held: the device's error words travel with the picture */
{
    hipLaunchKernelGGL(k_err_words, dim3(1), dim3(64), 0, st, u->e->d_err, u->e->hd_err);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(u->s->out_ev, st));
    return 0;
}
static int out_wait(SinkUser *u)                      /* mutex NOT held */
{
    if (hipEventSynchronize(u->s->out_ev) != hipSuccess) return -1;
    std::lock_guard<std::mutex> lk(u->e->mu);
    fold_errors(u->e);                                 /* (a later copy may have overwritten h_err meanwhile: it is as good or newer) */
    return 0;
}

int sink_fetch_begin(void *user, uint32_t slot)
{
    SinkUser *u = static_cast<SinkUser *>(user);
    StreamCtx *s = u->s;
    std::lock_guard<std::mutex> lk(u->e->mu);
    hipStream_t st;
    s->out_slot = -1;
    if (slot >= s->n_slots || out_begin(u, true, &st)) return -1;
    free_retired(s);
    if (!s->h_frame[slot] && hipHostMalloc((void **)&s->h_frame[slot], s->frame_bytes, hipHostMallocDefault) != hipSuccess) { s->h_frame[slot] = nullptr; return -1; }
    if (!s->hd_frame[slot] && hipHostGetDevicePointer((void **)&s->hd_frame[slot], s->h_frame[slot], 0) != hipSuccess) { s->hd_frame[slot] = nullptr; return -1; }
    /* the layout kernel writes the host mirror itself: 16-byte pieces over the link run at 49 GB/s (tools/probes/d2h_probe.hip: as fast as
     * whole rows, and as fast as the copy engine moves one large buffer), and a picture costs ONE launch on the compute queue instead
     * of a kernel, two copy-engine transfers and the hand-overs between the engines */
    hipLaunchKernelGGL(h264k::k_detile, dim3(512, 1), dim3(256), 0, st, s->d_frames + (size_t)slot * s->frame_bytes,
                       s->hd_frame[slot], s->wmb, s->hmb, (size_t)0, (size_t)0);
    if (hipGetLastError() != hipSuccess) return -1;
    if (out_end_locked(u, st)) return -1;
    s->out_slot = (int)slot;
    return 0;
}
uint8_t *sink_fetch_end(void *user)
{
    SinkUser *u = static_cast<SinkUser *>(user);
    StreamCtx *s = u->s;
    if (s->out_slot < 0 || out_wait(u)) return nullptr;
    return s->h_frame[s->out_slot];
}
uint8_t *sink_fetch(void *user, uint32_t slot) { return sink_fetch_begin(user, slot) ? nullptr : sink_fetch_end(user); }

uint32_t *sink_fetch_converted(void *user, uint32_t slot, int fmt)
{
    SinkUser *u = static_cast<SinkUser *>(user);
    StreamCtx *s = u->s;
    {
        std::lock_guard<std::mutex> lk(u->e->mu);
        hipStream_t st;
        if (slot >= s->n_slots || out_begin(u, true, &st)) return nullptr;
        const uint32_t w = s->wmb * 16, h = s->hmb * 16;
        const size_t bytes = (size_t)w * h * 4;
        if (!s->h_conv && hipHostMalloc((void **)&s->h_conv, bytes, hipHostMallocDefault) != hipSuccess) { s->h_conv = nullptr; return nullptr; }
        if (!s->hd_conv && hipHostGetDevicePointer((void **)&s->hd_conv, s->h_conv, 0) != hipSuccess) { s->hd_conv = nullptr; return nullptr; }
        hipLaunchKernelGGL(h264k::k_convert_tiles, dim3(1024, 1), dim3(256), 0, st,
                           s->d_frames + (size_t)slot * s->frame_bytes, s->hd_conv, s->wmb, s->hmb, fmt, (size_t)0, (size_t)0);
        if (hipGetLastError() != hipSuccess) return nullptr;
        if (out_end_locked(u, st)) return nullptr;
    }
    if (out_wait(u)) return nullptr;
    return s->h_conv;
}

void *sink_fetch_device(void *user, uint32_t slot, int fmt, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, void **stream)
{
    SinkUser *u = static_cast<SinkUser *>(user);
    StreamCtx *s = u->s;
    void *ret;
    {
        std::lock_guard<std::mutex> lk(u->e->mu);
        hipStream_t st;
        if (slot >= s->n_slots || out_begin(u, false, &st)) return nullptr;
        const uint32_t fw = s->wmb * 16, fh = s->hmb * 16;
        if (!w || !h || x0 + w > fw || y0 + h > fh) return nullptr;
        uint8_t *frame = s->d_frames + (size_t)slot * s->frame_bytes;
        /* frames are macroblock tiles in HBM: every picture that leaves is laid out by a kernel, the whole uncropped I420
         * frame by k_detile, everything else (window, conversion) by k_output */
        const size_t bytes = (size_t)fw * fh * 4;
        if (!s->d_conv && hipMalloc((void **)&s->d_conv, bytes) != hipSuccess) return nullptr;
        if (fmt == 3 && x0 == 0 && y0 == 0 && w == fw && h == fh) {
            hipLaunchKernelGGL(h264k::k_detile, dim3(512, 1), dim3(256), 0, u->e->stream, frame, reinterpret_cast<uint8_t *>(s->d_conv),
                               s->wmb, s->hmb, (size_t)0, (size_t)0);
        } else {
            const uint32_t n = fmt == 3 ? w * h * 3 / 8 : w * h / 4;
            hipLaunchKernelGGL(h264k::k_output, dim3((n + 255) / 256 < 2048 ? (n + 255) / 256 + 1 : 2048), dim3(256), 0, u->e->stream,
                               frame, reinterpret_cast<uint8_t *>(s->d_conv), fw, fh, fmt, x0, y0, w, h);
        }
        ret = s->d_conv;
        if (out_end_locked(u, st)) return nullptr;
        if (stream) *stream = st;
    }
    if (out_wait(u)) return nullptr;
    return ret;
}

void sink_close(void *user)
{
    SinkUser *u = static_cast<SinkUser *>(user);
    {
        std::lock_guard<std::mutex> lk(u->e->mu);
        hipSetDevice(u->e->device);
        reap_locked(u->e, true);
        hipStreamSynchronize(u->e->stream);
        if (u->e->out_stream) hipStreamSynchronize(u->e->out_stream);      /* (a pull that failed half way may have left its kernel behind) */
        stream_release(u->s);
        auto &v = u->e->streams;
        v.erase(std::remove(v.begin(), v.end(), u->s), v.end());
    }
    delete u->s;
    delete u;
}

} // namespace

/* ================================================================== C interface */
extern "C" {

int eng_attach(JobSink *sink)
{
    Engine *e = engine_get();
    if (!e) return -1;
    SinkUser *u = new SinkUser{ e, new StreamCtx() };
    {
        std::lock_guard<std::mutex> lk(e->mu);
        u->s->group = (int)e->group_rr++;             /* taken modulo the number of light lanes */
        e->streams.push_back(u->s);
    }
    sink->user = u;
    sink->configure = sink_configure;
    sink->acquire = sink_acquire;
    sink->submit = sink_submit;
    sink->fetch = sink_fetch;
    sink->fetch_begin = sink_fetch_begin;
    sink->fetch_end = sink_fetch_end;
    sink->fetch_converted = sink_fetch_converted;
    sink->fetch_device = sink_fetch_device;
    sink->close = sink_close;
    sink->errors = sink_errors;
    return 0;
}

void eng_convert_host(int fmt, uint32_t width, uint32_t height, const uint8_t *data, uint32_t *out)
{
    Engine *e = engine_get();
    if (!e) {
        fprintf(stderr, "h264bsd-mi355x: h264bsdConvertTo*: no usable HIP device (this library has no CPU pixel path)\n");
        return;
    }
    std::lock_guard<std::mutex> lk(e->mu);
    const size_t in_b = (size_t)width * height * 3 / 2, out_b = (size_t)width * height * 4;
    if (hipSetDevice(e->device) != hipSuccess) return;
    if (out_b > e->conv_cap) {
        if (e->conv_in) hipFree(e->conv_in);
        if (e->conv_out) hipFree(e->conv_out);
        e->conv_in = nullptr; e->conv_out = nullptr; e->conv_cap = 0;
        if (hipMalloc((void **)&e->conv_in, in_b + 64) != hipSuccess || hipMalloc((void **)&e->conv_out, out_b) != hipSuccess) return;
        e->conv_cap = out_b;
    }
    if (hipMemcpyAsync(e->conv_in, data, in_b, hipMemcpyHostToDevice, e->stream) != hipSuccess) return;
    hipLaunchKernelGGL(h264k::k_convert, dim3(1024, 1), dim3(256), 0, e->stream, e->conv_in, e->conv_out, width, height, fmt,
                       (size_t)0, (size_t)0, 0);
    if (hipMemcpyAsync(out, e->conv_out, out_b, hipMemcpyDeviceToHost, e->stream) != hipSuccess) return;
    hipStreamSynchronize(e->stream);
}

int h264bsdmiDeviceCount(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

/* Decoder instances and replay sets created by the calling thread from now on live on `device`; the first call also
 * sets the process default (threads that never call it).  Existing instances stay where they are. */
int h264bsdmiSetDevice(int device)
{
    int n = 0;
    if (device < 0 || device >= MAX_DEVICES || hipGetDeviceCount(&n) != hipSuccess || device >= n) return -1;
    tl_device = device;
    std::lock_guard<std::mutex> lk(g_engine_mu);
    if (g_default_device < 0) g_default_device = device;
    return 0;
}

/* the device (and with it the NUMA node: eng_device_cpus) a decoder instance lives on */
int eng_sink_device(const JobSink *sink)
{
    if (!sink || !sink->user || sink->submit != sink_submit) return -1;      /* capture-mode instances have another sink */
    return static_cast<const SinkUser *>(sink->user)->e->device;
}

/* CPUs close to a device: /sys/bus/pci/devices/<bus id>/local_cpulist ("0-63,128-191"), for pinning parser threads to
 * the NUMA node of the GPU their streams feed.  Returns the number of CPUs written to cpus[]. */
int eng_device_cpus(int device, int *cpus, int max)
{
    char bus[64], path[160], line[4096];
    if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device) != hipSuccess) return 0;
    for (char *c = bus; *c; c++) if (*c >= 'A' && *c <= 'F') *c = (char)(*c - 'A' + 'a');
    snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/local_cpulist", bus);
    FILE *f = fopen(path, "r");
    if (!f) return 0;
    int n = 0;
    if (fgets(line, sizeof(line), f)) {
        for (char *p = line; *p && n < max;) {
            char *end;
            long a = strtol(p, &end, 10), b = a;
            if (end == p) break;
            if (*end == '-') { p = end + 1; b = strtol(p, &end, 10); }
            for (long c = a; c <= b && n < max; c++) cpus[n++] = (int)c;
            p = *end == ',' ? end + 1 : end;
            if (*end != ',' ) break;
        }
    }
    fclose(f);
    return n;
}

unsigned h264bsdmiDeviceErrors(void)
{
    unsigned all = 0;
    for (int d = 0; d < MAX_DEVICES; d++) {
        Engine *e;
        { std::lock_guard<std::mutex> lk(g_engine_mu); e = g_engines[d]; }
        if (!e) continue;
        std::lock_guard<std::mutex> lk(e->mu);
        if (hipSetDevice(e->device) != hipSuccess || poll_errors(e)) return 0xFFFFFFFFu;
        all |= e->errors;
    }
    return all;
}

/* test harness (bench library): tripwire events of all devices so far, after waiting for the devices — monotonic, unlike the
 * sticky bits of h264bsdmiDeviceErrors(): a test asserts that its own pictures added none */
unsigned h264bsdmiDebugDeviceErrorEvents(void)
{
    unsigned all = 0;
    for (int d = 0; d < MAX_DEVICES; d++) {
        Engine *e;
        { std::lock_guard<std::mutex> lk(g_engine_mu); e = g_engines[d]; }
        if (!e) continue;
        std::lock_guard<std::mutex> lk(e->mu);
        if (hipSetDevice(e->device) != hipSuccess || poll_errors(e)) return 0xFFFFFFFFu;
        all += e->error_events;
    }
    return all;
}

#ifdef H264K_INTER_PROFILE
/* profiling build only: the 24 64-bit counters behind the device error word (k_recon_inter's cycle accounting), read and zeroed */
int h264bsdmiDebugReadCounters(unsigned long long *out)
{
    Engine *e = engine_get();
    if (!e) return -1;
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, reinterpret_cast<uint8_t *>(e->d_err) + 64, 192, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemset(reinterpret_cast<uint8_t *>(e->d_err) + 64, 0, 192));
    HIP_TRY(hipDeviceSynchronize());
    return 0;
}
#endif

static int flush_all(bool wait)
{
    int rc = 0, any = 0;
    for (int d = 0; d < MAX_DEVICES; d++) {
        Engine *e;
        { std::lock_guard<std::mutex> lk(g_engine_mu); e = g_engines[d]; }
        if (!e) continue;
        any = 1;
        std::lock_guard<std::mutex> lk(e->mu);
        if (flush_locked(e, wait)) rc = -1;
    }
    if (!any && !engine_get()) return -1;
    return rc;
}

int h264bsdmiFlush(void) { return flush_all(true); }

int h264bsdmiFlushAsync(void) { return flush_all(false); }

/* ------------------------------------------------------------------ replay sets */
struct h264bsdmi_replay {
    Engine *e;
    uint32_t n_pics, n_streams, n_slots, wmb, hmb, frame_bytes;
    size_t blob_stride;               /* bytes of all blobs of one stream (256-aligned) */
    unsigned long long job_bytes;     /* sum of the blob sizes of one stream */
    uint8_t *d_blobs;                 /* n_streams * blob_stride */
    uint8_t *d_frames;                /* n_streams * n_slots * frame_bytes */
    uint8_t *d_dbk;                   /* n_streams * n_mbs * 32 */
    FrameDesc *d_desc;                /* n_pics * n_streams */
    uint32_t *d_conv;                 /* n_streams * w*h (lazy) */
    uint8_t *d_planar = nullptr;      /* one frame, planar (h264bsdmiReplayFetch) */
    unsigned long long *d_sums;
    std::vector<TickShape> shapes;
    std::vector<uint8_t> cur_slot;
    std::vector<TickTimers> timers;
    uint32_t timed_first, timed_count;
    hipEvent_t ev_begin, ev_end, gdone_any = nullptr;
    uint32_t launches[5];
    unsigned stages;
    uint32_t n_groups;
    hipStream_t gstream[8];
    hipEvent_t gdone[8];
    bool overlap_dbk = true;
    unsigned timed_mask = 31u;
    /* desynchronised sets with heavy lanes (h264bsdmiReplayCreateDesync, lanes > 0): a static launch schedule */
    struct Launch { size_t first; TickShape shape; int lane; std::vector<int> waits; int record_ev; bool light; };
    std::vector<Launch> sched;
    std::vector<hipEvent_t> sched_ev;
    static constexpr int MAX_LANES = 72;  /* lanes 0..n_light-1: one per stream group (light pictures), then the heavy lanes */
    hipStream_t lanes[MAX_LANES] = {};
    SideLane lane_side[MAX_LANES];        /* k_dbk next to the reconstruction kernels, per light lane */
    uint32_t n_lanes = 0, n_light = 0;
    /* HIP streams of earlier schedules of this set, reused by the next one (normal priority: light lanes and stream groups; highest: heavy
     * lanes; side-stream pairs).  A process that creates and destroys a dozen streams per schedule falls off the runtime's stream cliff after
     * a few of them (a 12-lane schedule then takes seconds per lap, the same schedule in a fresh process 0.13 s): nothing is destroyed before the set is. */
    std::vector<hipStream_t> pool_normal, pool_high;
    std::vector<SideLane> pool_side;
    void retire_streams()
    {
        for (uint32_t k = 0; k < (uint32_t)MAX_LANES; k++) {
            if (lanes[k]) { (k < n_light ? pool_normal : pool_high).push_back(lanes[k]); lanes[k] = nullptr; }
            if (lane_side[k].stream) { pool_side.push_back(lane_side[k]); lane_side[k] = SideLane(); }
        }
    }
    bool take_stream(hipStream_t *st, bool high, int prio)
    {
        std::vector<hipStream_t> &pool = high ? pool_high : pool_normal;
        if (!pool.empty()) { *st = pool.back(); pool.pop_back(); return true; }
        return (high ? hipStreamCreateWithPriority(st, hipStreamNonBlocking, prio) : hipStreamCreateWithFlags(st, hipStreamNonBlocking)) == hipSuccess;
    }
    bool take_side(SideLane *sl)
    {
        if (!pool_side.empty()) { *sl = pool_side.back(); pool_side.pop_back(); return true; }
        return sl->create(0, false);
    }
    std::vector<uint32_t> offsets;    /* first picture of every stream */
    /* what a schedule is built from (replay_schedule: at creation and again for every h264bsdmiReplayReschedule) */
    std::vector<FjHeader> heads;      /* the headers of the n_pics jobs (host copies) */
    std::vector<size_t> blob_off;     /* where job p lies inside a stream's blobs */
    size_t frames_per_stream = 0, dbk_half = 0, dbk_stride = 0;
    /* config 3 ("ARGB conversion on-GPU"): colour conversion of every produced picture inside the run, timed */
    int convert_fmt = -1;
    std::vector<hipEvent_t> cev;      /* 2 per tick */
    /* ... hosted by the NEXT tick's k_frame_dbk where that is possible (kernels/convert.hip.h, conv_drain): descriptors with the
     * conversion of the stream's previous picture written in, which ticks host */
    FrameDesc *d_desc_conv = nullptr;
    std::vector<uint8_t> hosted;      /* tick i converts the pictures of tick i - 1 while it filters its own */
    std::vector<uint8_t> cev_on;      /* tick i was followed by a stand-alone conversion launch in the last run */
    bool host_convert = true, convert_trailing = true;
    uint32_t conv_waves = 0;
};

h264bsdmi_replay *h264bsdmiReplayCreateSched(const u8 *const *blobs, const u32 *bytes, u32 n_pics, u32 n_streams,
                                             const u32 *offsets, u32 heavy_lanes, u32 heavy_delay, u32 groups);

/* Descriptors and launch schedule of a replay set for the offsets in r->offsets: lock-step / staggered / common ticks
 * (heavy_lanes == 0, groups <= 1: tick i = picture (i + offset) mod n_pics of every stream) or the static schedule of
 * stream groups and heavy lanes (h264bsdmiReplayCreateDesync).  Called at creation and by h264bsdmiReplayReschedule,
 * which has torn the previous schedule down. */
static bool replay_schedule(h264bsdmi_replay *r, u32 heavy_lanes, u32 heavy_delay, u32 groups)
{
    Engine *e = r->e;
    const u32 n_pics = r->n_pics, n_streams = r->n_streams;
    const size_t total = r->blob_stride, frames_per_stream = r->frames_per_stream, dbk_stride = r->dbk_stride;
    const std::vector<size_t> &offs = r->blob_off;
    auto blob_of = [&](u32 p) { return reinterpret_cast<const uint8_t *>(&r->heads[p]); };     /* make_desc reads the header only */
    bool ok = true;
    r->shapes.assign(n_pics, TickShape());
    for (u32 i = 0; i < n_pics; i++) {
        TickShape s0;
        FrameDesc tmp;
        make_desc(tmp, blob_of(i), nullptr, nullptr, 0, nullptr, &s0, nullptr);
        s0.n_frames = n_streams;
        r->shapes[i] = s0;
    }
    if (ok) {
        std::vector<FrameDesc> descs((size_t)n_pics * n_streams);
        auto desc_of = [&](FrameDesc &d, u32 s, u32 p, TickShape *shape, u32 tick = 0) {
            make_desc(d, blob_of(p), r->d_blobs + (size_t)s * total + offs[p], r->d_frames + (size_t)s * frames_per_stream,
                      r->frame_bytes, r->d_dbk + (size_t)s * dbk_stride, shape, e->d_err);
        };
        if (!heavy_lanes && groups <= 1) {
            for (u32 i = 0; i < n_pics; i++) {
                TickShape shape;                          /* a tick is as large as the largest of its pictures */
                for (u32 s = 0; s < n_streams; s++) desc_of(descs[(size_t)i * n_streams + s], s, (i + r->offsets[s]) % n_pics, &shape, i);
                r->shapes[i] = shape;
            }
        } else {
            /* static schedule: every group's light ticks on its own lane, heavy pictures round-robin on the heavy lanes */
            std::vector<u32> done(n_streams, 0), ready_at(n_streams, 0);
            std::vector<int> last_ev(n_streams, -1);     /* event of the heavy launch a stream's previous picture ran in */
            size_t n_desc = 0;
            u32 left = n_streams, heavy_count = 0;
            auto is_heavy = [&](u32 p) { const FjHeader *h = reinterpret_cast<const FjHeader *>(blob_of(p)); return heavy_lanes && h->n_intra * 4u > h->n_mbs; };      /* (no heavy lanes: heavy pictures stay in their group's tick) */
            /* Cost-affine groups: a group's tick lasts as long as its slowest picture, so streams whose next pictures cost
             * about the same belong together.  Every REGROUP rounds the streams are sorted by the estimated per-picture
             * kernel time of their next REGROUP pictures (from the job headers: intra and filtered macroblock counts) and
             * dealt to the groups in that order; a stream that changes groups makes its new lane wait for the event its
             * old lane recorded after the last round before the regrouping. */
#ifndef REGROUP_ROUNDS
#define REGROUP_ROUNDS 32      /* measured: 4: 681, 8: 664, 16: 699, 32: 709-733 M MB/s (8-9 groups); every regrouping costs cross-lane waits */
#endif
            constexpr u32 REGROUP = REGROUP_ROUNDS;
            std::vector<std::vector<u32>> members(groups);
            std::vector<u32> group_of(n_streams, 0);
            std::vector<int> pre_regroup_ev(groups, -1);
            auto upcoming_cost = [&](u32 s) {
                uint64_t c = 0;
                for (u32 i = 0; i < REGROUP && done[s] + i < n_pics; i++) {
                    const FjHeader *h = &r->heads[(r->offsets[s] + done[s] + i) % n_pics];
                    c += 11u * h->n_intra + 4u * h->n_dbk;          /* ~0.55 us per intra macroblock, ~0.2 us per filtered one */
                }
                return c;
            };
            auto new_event = [&]() { r->sched_ev.push_back(nullptr); return (int)r->sched_ev.size() - 1; };
            for (u32 t = 0; left && t < 16u * n_pics; t++) {
                /* one heavy launch per round for the heavy pictures of all groups: it waits for the light launch of
                 * every group it takes a stream from (the previous picture of that stream ran there or earlier) */
                std::vector<u32> hs;
                std::vector<int> hwaits;
                if (t % REGROUP == 0) {
                    std::vector<std::pair<uint64_t, u32>> order;
                    for (u32 s = 0; s < n_streams; s++) if (done[s] < n_pics) order.emplace_back(upcoming_cost(s), s);
                    std::sort(order.begin(), order.end());
                    for (auto &m : members) m.clear();
                    for (size_t i = 0; i < order.size(); i++) {
                        const u32 s = order[i].second, g = (u32)(i * groups / order.size());
                        if (t && g != group_of[s] && last_ev[s] < 0) last_ev[s] = pre_regroup_ev[group_of[s]];
                        group_of[s] = g;
                        members[g].push_back(s);
                    }
                    for (auto &m : members) std::sort(m.begin(), m.end());
                }
                const bool before_regroup = (t + 1) % REGROUP == 0;
                for (u32 g = 0; g < groups; g++) {
                    h264bsdmi_replay::Launch light{ n_desc, TickShape(), (int)g, {}, -1, true };
                    bool group_has_heavy = false;
                    for (u32 s : members[g]) {
                        if (done[s] >= n_pics || ready_at[s] > t) continue;
                        const u32 p = (r->offsets[s] + done[s]) % n_pics;
                        if (is_heavy(p)) { hs.push_back(s); group_has_heavy = true; continue; }
                        if (last_ev[s] >= 0) {               /* rejoining after a heavy picture */
                            if (std::find(light.waits.begin(), light.waits.end(), last_ev[s]) == light.waits.end()) light.waits.push_back(last_ev[s]);
                            last_ev[s] = -1;
                        }
                        desc_of(descs[n_desc++], s, p, &light.shape);
                        if (++done[s] == n_pics) left--;
                    }
                    const bool have_light = light.shape.n_frames != 0;
                    if (have_light) {
                        if (group_has_heavy || before_regroup) light.record_ev = new_event();
                        if (group_has_heavy) hwaits.push_back(light.record_ev);
                        if (before_regroup) pre_regroup_ev[g] = light.record_ev;
                        light.shape.dbk_waves = LANE_DBK_WAVES;
                        r->sched.push_back(light);
                    } else {
                        if (group_has_heavy) hwaits.push_back(-2 - (int)g);        /* "everything enqueued on light lane g so far" */
                        if (before_regroup) {                                      /* an empty launch: only the event */
                            light.record_ev = pre_regroup_ev[g] = new_event();
                            r->sched.push_back(light);
                        }
                    }
                }
                if (!hs.empty()) {
                    h264bsdmi_replay::Launch heavy{ n_desc, TickShape(), (int)(groups + heavy_count++ % heavy_lanes), hwaits, -1, false };
                    for (u32 s : hs) {
                        if (last_ev[s] >= 0 && std::find(heavy.waits.begin(), heavy.waits.end(), last_ev[s]) == heavy.waits.end()) heavy.waits.push_back(last_ev[s]);
                        desc_of(descs[n_desc++], s, (r->offsets[s] + done[s]) % n_pics, &heavy.shape);
                        if (++done[s] == n_pics) left--;
                        ready_at[s] = t + 1 + heavy_delay;
                    }
                    heavy.record_ev = new_event();
                    for (u32 s : hs) last_ev[s] = heavy.record_ev;
                    r->sched.push_back(heavy);
                }
            }
            if (left || n_desc != descs.size()) ok = false;
            r->n_light = groups;
            r->n_lanes = groups + heavy_lanes;
            /* the heavy pictures' workgroups need a whole compute unit each: highest priority (measured: no
             * difference on this runtime, kept because it states the intent) */
            int prio_least = 0, prio_greatest = 0;
            if (hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest) != hipSuccess) prio_greatest = 0;
            for (u32 k = 0; ok && k < r->n_lanes; k++) {
                if (k < groups) {
                    ok = r->take_stream(&r->lanes[k], false, 0);
                    if (ok && groups <= 2)        /* with more groups the other groups are the overlap, and busy HIP streams are scarce (Lane, above) */
                        ok = r->take_side(&r->lane_side[k]);
                } else ok = r->take_stream(&r->lanes[k], true, prio_greatest);
            }
            for (auto &ev : r->sched_ev) if (ok) ok = hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess;
        }
        if (ok) ok = hipMemcpyAsync(r->d_desc, descs.data(), descs.size() * sizeof(FrameDesc), hipMemcpyHostToDevice, e->stream) == hipSuccess &&
                     hipStreamSynchronize(e->stream) == hipSuccess;
    }
    return ok;
}

h264bsdmi_replay *h264bsdmiReplayCreate(const u8 *const *blobs, const u32 *bytes, u32 n_pics, u32 n_streams)
{
    return h264bsdmiReplayCreateDesync(blobs, bytes, n_pics, n_streams, nullptr, 0, 0);
}

/* odd_offset != 0: the "staggered" variant of SURVEY.md §8d config 4 — odd-numbered streams run picture
 * (i + odd_offset) mod n_pics in tick i (odd_offset must be the index of an IDR picture, so that both the
 * start and the wrap-around are clean decoder starts); every tick then mixes two different pictures */
h264bsdmi_replay *h264bsdmiReplayCreateStaggered(const u8 *const *blobs, const u32 *bytes, u32 n_pics, u32 n_streams, u32 odd_offset)
{
    if (odd_offset >= n_pics) return nullptr;
    std::vector<u32> offs(n_streams, 0);
    for (u32 s = 1; s < n_streams; s += 2) offs[s] = odd_offset;
    return h264bsdmiReplayCreateDesync(blobs, bytes, n_pics, n_streams, offs.data(), 0, 0);
}

/* Streams that are NOT in step: stream s starts at picture offsets[s] (nullptr = all 0) and runs n_pics pictures,
 * wrapping around (picture 0 must be an IDR picture).  heavy_lanes == 0: tick i holds picture (i + offsets[s]) mod
 * n_pics of every stream — a tick then lasts as long as its slowest picture.  heavy_lanes > 0: a static schedule of
 * what a scheduler achieves that keeps light pictures from waiting for heavy ones:
 *   - the streams are split into `groups` groups (stream s -> group s % groups), every group runs its own ticks on its
 *     own HIP stream ("light lane"): a group's tick lasts as long as ITS slowest picture, and the workgroups of the
 *     other groups fill the compute units it leaves idle (tail kernels are one workgroup per picture);
 *   - pictures that are mostly intra-coded ("heavy", more than a quarter of their macroblocks) leave their group's
 *     tick and run on one of heavy_lanes extra HIP streams; their stream of pictures rejoins its group heavy_delay
 *     ticks later (an event makes the group's tick wait if the heavy picture is not finished by then). */
h264bsdmi_replay *h264bsdmiReplayCreateDesync(const u8 *const *blobs, const u32 *bytes, u32 n_pics, u32 n_streams,
                                              const u32 *offsets, u32 heavy_lanes, u32 heavy_delay)
{
    return h264bsdmiReplayCreateSched(blobs, bytes, n_pics, n_streams, offsets, heavy_lanes, heavy_delay, 1);
}

h264bsdmi_replay *h264bsdmiReplayCreateSched(const u8 *const *blobs, const u32 *bytes, u32 n_pics, u32 n_streams,
                                             const u32 *offsets, u32 heavy_lanes, u32 heavy_delay, u32 groups)
{
    if (groups < 1) groups = 1;
    if (groups > 16 || groups > n_streams) return nullptr;
    if (heavy_lanes + groups > (u32)h264bsdmi_replay::MAX_LANES) return nullptr;
    for (u32 s = 0; offsets && s < n_streams; s++) if (offsets[s] >= n_pics) return nullptr;
    Engine *e = engine_get();
    if (!e || !n_pics || !n_streams) {
        if (!e) fprintf(stderr, "h264bsd-mi355x: h264bsdmiReplayCreate: no usable HIP device\n");
        return nullptr;
    }
    std::lock_guard<std::mutex> lk(e->mu);
    if (hipSetDevice(e->device) != hipSuccess) return nullptr;
    h264bsdmi_replay *r = new h264bsdmi_replay();
    r->e = e; r->n_pics = n_pics; r->n_streams = n_streams;
    r->offsets.assign(n_streams, 0);
    if (offsets) r->offsets.assign(offsets, offsets + n_streams);
    const FjHeader *h0 = reinterpret_cast<const FjHeader *>(blobs[0]);
    r->wmb = h0->width_mbs; r->hmb = h0->height_mbs; r->n_slots = h0->n_slots;
    r->frame_bytes = fj_frame_bytes(r->wmb, r->hmb);
    std::vector<size_t> offs(n_pics);
    size_t total = 0;
    r->job_bytes = 0;
    for (u32 i = 0; i < n_pics; i++) { offs[i] = total; total += ((size_t)bytes[i] + 255u) & ~(size_t)255u; r->job_bytes += bytes[i]; }
    r->blob_stride = total;
    r->blob_off = offs;
    r->d_blobs = nullptr; r->d_frames = nullptr; r->d_desc = nullptr; r->d_conv = nullptr; r->d_sums = nullptr; r->d_dbk = nullptr;
    const size_t frames_per_stream = (size_t)r->n_slots * r->frame_bytes;
    const size_t dbk_half = (DBK_SCRATCH_BYTES(h0->n_mbs) + 255) & ~(size_t)255, dbk_stride = dbk_half;
    r->frames_per_stream = frames_per_stream; r->dbk_half = dbk_half; r->dbk_stride = dbk_stride;
    bool ok = hipMalloc((void **)&r->d_blobs, total * n_streams) == hipSuccess &&
              hipMalloc((void **)&r->d_frames, frames_per_stream * n_streams + 256) == hipSuccess &&
              hipMalloc((void **)&r->d_desc, sizeof(FrameDesc) * (size_t)n_pics * n_streams) == hipSuccess &&
              hipMalloc((void **)&r->d_sums, sizeof(unsigned long long) * n_streams) == hipSuccess &&
              hipMalloc((void **)&r->d_dbk, (size_t)n_streams * dbk_stride) == hipSuccess;
    if (ok) ok = hipMemsetAsync(r->d_dbk, 0, (size_t)n_streams * dbk_stride, e->stream) == hipSuccess;
    if (ok) ok = hipMemsetAsync(r->d_frames, 0, frames_per_stream * n_streams + 256, e->stream) == hipSuccess;
    /* stream 0 from the host, the other copies device-to-device: every stream owns private jobs */
    for (u32 i = 0; ok && i < n_pics; i++) {
        ok = hipMemcpyAsync(r->d_blobs + offs[i], blobs[i], bytes[i], hipMemcpyHostToDevice, e->stream) == hipSuccess;
        const FjHeader *h = reinterpret_cast<const FjHeader *>(blobs[i]);
        if (h->width_mbs != r->wmb || h->height_mbs != r->hmb || h->n_slots != r->n_slots) ok = false;
        r->heads.push_back(*h);
        r->cur_slot.push_back(h->cur_slot);
    }
    if (ok) ok = hipStreamSynchronize(e->stream) == hipSuccess;
    for (u32 s = 1; ok && s < n_streams; s++)
        ok = hipMemcpyAsync(r->d_blobs + (size_t)s * total, r->d_blobs, total, hipMemcpyDeviceToDevice, e->stream) == hipSuccess;
    if (ok) ok = replay_schedule(r, heavy_lanes, heavy_delay, groups);
    r->timers.resize(n_pics);
    for (auto &t : r->timers) {
        for (auto &ev : t.ev) if (ok) ok = hipEventCreate(&ev) == hipSuccess;
        for (auto &ev : t.sev) if (ok) ok = hipEventCreate(&ev) == hipSuccess;
    }
    if (ok) ok = hipEventCreate(&r->ev_begin) == hipSuccess && hipEventCreate(&r->ev_end) == hipSuccess &&
                 hipEventCreateWithFlags(&r->gdone_any, hipEventDisableTiming) == hipSuccess;
    r->timed_first = r->timed_count = 0;
    r->stages = 7u;
    r->n_groups = 1;
    for (int g = 0; g < 8; g++) { r->gstream[g] = nullptr; r->gdone[g] = nullptr; }
    if (!ok) {
        fprintf(stderr, "h264bsd-mi355x: h264bsdmiReplayCreate failed (%s)\n", hipGetErrorString(hipGetLastError()));
        if (r->d_blobs) hipFree(r->d_blobs);
        if (r->d_frames) hipFree(r->d_frames);
        if (r->d_desc) hipFree(r->d_desc);
        if (r->d_sums) hipFree(r->d_sums);
        if (r->d_dbk) hipFree(r->d_dbk);
        delete r;
        return nullptr;
    }
    return r;
}

void h264bsdmiReplayDestroy(h264bsdmi_replay *r)
{
    if (!r) return;
    std::lock_guard<std::mutex> lk(r->e->mu);
    hipSetDevice(r->e->device);
    hipStreamSynchronize(r->e->stream);
    hipFree(r->d_blobs); hipFree(r->d_frames); hipFree(r->d_desc); hipFree(r->d_sums); hipFree(r->d_dbk);
    if (r->d_conv) hipFree(r->d_conv);
    if (r->d_desc_conv) hipFree(r->d_desc_conv);
    if (r->d_planar) hipFree(r->d_planar);
    for (auto &t : r->timers) for (auto &ev : t.ev) hipEventDestroy(ev);
    hipEventDestroy(r->ev_begin); hipEventDestroy(r->ev_end); if (r->gdone_any) hipEventDestroy(r->gdone_any);
    for (int g = 0; g < 8; g++) { if (r->gstream[g]) { tickets_release(r->gstream[g]); hipStreamDestroy(r->gstream[g]); } if (r->gdone[g]) hipEventDestroy(r->gdone[g]); }
    for (auto &ev : r->sched_ev) if (ev) hipEventDestroy(ev);
    for (auto &ev : r->cev) hipEventDestroy(ev);
    r->retire_streams();
    for (auto *pool : { &r->pool_normal, &r->pool_high }) for (auto &st : *pool) { tickets_release(st); hipStreamDestroy(st); }
    for (auto &sl : r->pool_side) sl.destroy();
    delete r;
}

/* The same resident jobs and frame buffers under another schedule (other first pictures, heavy lanes, stream groups): what
 * a second h264bsdmiReplayCreate* would build, without allocating and uploading 20 GB again.  Frame buffers and deblocking
 * scratch start from zero like those of a new set.  0 = ok; after a failure the set can only be destroyed. */
int h264bsdmiReplayReschedule(h264bsdmi_replay *r, const u32 *offsets, u32 heavy_lanes, u32 heavy_delay, u32 groups)
{
    if (!r) return -1;
    if (groups < 1) groups = 1;
    if (groups > 16 || groups > r->n_streams || heavy_lanes + groups > (u32)h264bsdmi_replay::MAX_LANES) return -1;
    for (u32 s = 0; offsets && s < r->n_streams; s++) if (offsets[s] >= r->n_pics) return -1;
    Engine *e = r->e;
    std::lock_guard<std::mutex> lk(e->mu);
    HIP_TRY(hipSetDevice(e->device));
    /* everything the old schedule launched has to be over before its streams and events go */
    for (auto &st : r->lanes) if (st) HIP_TRY(hipStreamSynchronize(st));
    for (int g = 0; g < 8; g++) if (r->gstream[g]) HIP_TRY(hipStreamSynchronize(r->gstream[g]));
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (poll_errors(e)) return -1;
    for (auto &ev : r->sched_ev) if (ev) hipEventDestroy(ev);
    r->sched_ev.clear(); r->sched.clear();
    r->retire_streams();                               /* (kept for the next schedule: h264bsdmi_replay::pool_*) */
    for (int g = 0; g < 8; g++) if (r->gstream[g]) { r->pool_normal.push_back(r->gstream[g]); r->gstream[g] = nullptr; }      /* h264bsdmiReplaySetGroups takes them back */
    r->n_lanes = r->n_light = 0;
    r->n_groups = 1;                                  /* (h264bsdmiReplaySetGroups: a property of the schedule it was set for) */
    r->convert_fmt = -1; r->timed_mask = 31u; r->stages = 7u;
    r->offsets.assign(r->n_streams, 0);
    if (offsets) r->offsets.assign(offsets, offsets + r->n_streams);
    HIP_TRY(hipMemsetAsync(r->d_dbk, 0, (size_t)r->n_streams * r->dbk_stride, e->stream));
    HIP_TRY(hipMemsetAsync(r->d_frames, 0, r->frames_per_stream * r->n_streams + 256, e->stream));
    if (!replay_schedule(r, heavy_lanes, heavy_delay, groups)) return -1;
    r->timed_first = r->timed_count = 0;
    return 0;
}

int h264bsdmiReplayRun(h264bsdmi_replay *r, u32 first, u32 count)
{
    if (!r || first + count > r->n_pics) return -1;
    std::lock_guard<std::mutex> lk(r->e->mu);
    HIP_TRY(hipSetDevice(r->e->device));
    r->timed_first = first; r->timed_count = count;
    for (auto &l : r->launches) l = 0;
    HIP_TRY(hipEventRecord(r->ev_begin, r->e->stream));
    if (!r->sched.empty()) {
        /* desynchronised set with lanes: one whole lap of the static schedule (first / count are ignored) */
        r->timed_count = 0;
        for (u32 k = 0; k < r->n_lanes; k++) HIP_TRY(hipStreamWaitEvent(r->lanes[k], r->ev_begin, 0));   /* the previous lap is complete */
        std::vector<hipEvent_t> lane_mark(r->n_light, nullptr);
        for (const auto &l : r->sched) {
            hipStream_t st = r->lanes[l.lane];
            for (int w : l.waits) {
                if (w >= 0) HIP_TRY(hipStreamWaitEvent(st, r->sched_ev[w], 0));
                else {                                   /* -2 - g: everything enqueued on light lane g so far */
                    HIP_TRY(hipEventRecord(r->gdone_any, r->lanes[-2 - w]));
                    HIP_TRY(hipStreamWaitEvent(st, r->gdone_any, 0));
                }
            }
            if (l.shape.n_frames && launch_tick(st, r->d_desc + l.first, [&] { TickShape sh = l.shape; sh.load = r->n_streams; return sh; }(), nullptr, r->launches, r->stages,
                                                (l.light && r->overlap_dbk && !(r->stages & 8u) && r->lane_side[l.lane].stream) ? &r->lane_side[l.lane] : nullptr)) return -1;
            if (l.record_ev >= 0) HIP_TRY(hipEventRecord(r->sched_ev[l.record_ev], st));
        }
        for (u32 k = 0; k < r->n_lanes; k++) {               /* the lap ends when every lane has drained */
            HIP_TRY(hipEventRecord(r->gdone_any, r->lanes[k]));
            HIP_TRY(hipStreamWaitEvent(r->e->stream, r->gdone_any, 0));
        }
    } else if (r->n_groups <= 1) {
        for (u32 i = first; i < first + count; i++) { r->timers[i].on = true; r->timers[i].mask = r->timed_mask; }
        r->cev_on.assign(r->n_pics, 0);
        for (u32 i = first; i < first + count; i++) {
            /* config 3: the pictures of tick i - 1 are converted by tick i's k_frame_dbk workgroups where the schedule allows it */
            const bool host = r->convert_fmt >= 0 && r->host_convert && i > first && r->hosted[i];
            TickShape shape = r->shapes[i];
            shape.conv = host; shape.conv_waves = r->conv_waves;
            if (launch_tick(r->e->stream, (host ? r->d_desc_conv : r->d_desc) + (size_t)i * r->n_streams, shape, &r->timers[i], r->launches, r->stages, (r->overlap_dbk && !(r->stages & 8u)) ? &r->e->side : nullptr, r->e->tail_prof)) return -1;
            const bool next_hosts = r->convert_fmt >= 0 && r->host_convert && i + 1 < first + count && r->hosted[i + 1];
            if (r->convert_fmt >= 0 && !next_hosts && (r->convert_trailing || i + 1 < first + count)) {
                /* the picture every stream has just produced, converted where it lies (tiles -> packed 32-bit pixels) */
                const uint32_t w = r->wmb * 16, h = r->hmb * 16;
                HIP_TRY(hipEventRecord(r->cev[2 * i], r->e->stream));
                hipLaunchKernelGGL(h264k::k_convert_tiles, CONVERT_GRID(r->n_streams), dim3(256), 0, r->e->stream,
                                   r->d_frames + (size_t)r->cur_slot[i] * r->frame_bytes, r->d_conv, r->wmb, r->hmb, r->convert_fmt,
                                   (size_t)r->n_slots * r->frame_bytes, (size_t)w * h);
                HIP_TRY(hipEventRecord(r->cev[2 * i + 1], r->e->stream));
                r->cev_on[i] = 1;
            }
        }
    } else {
        /* stream groups on separate HIP streams: the latency-bound per-picture tail of one group overlaps
         * with the throughput-bound inter reconstruction of another (pictures of different streams are
         * independent; every group still runs its own pictures strictly in order) */
        const u32 G = r->n_groups, per = (r->n_streams + G - 1) / G;
        for (u32 g = 0; g < G; g++) HIP_TRY(hipStreamWaitEvent(r->gstream[g], r->ev_begin, 0));
        for (u32 i = first; i < first + count; i++) {
            for (u32 g = 0; g < G; g++) {
                const u32 s0 = g * per, s1 = std::min(r->n_streams, s0 + per);
                if (s0 >= s1) continue;
                TickShape sh = r->shapes[i];
                sh.n_frames = s1 - s0;
                sh.load = r->n_streams;
                TickTimers &tt = r->timers[(size_t)g * r->n_pics + i];
                tt.on = true; tt.mask = r->timed_mask;
                /* (making the groups take turns at the list-driven kernels — a ring of events — works as designed in the kernel
                 * trace and loses: docs/EXPERIMENTS.md) */
                if (i == first && g > 0) HIP_TRY(hipStreamWaitEvent(r->gstream[g], r->timers[(size_t)(g - 1) * r->n_pics + i].ev[3], 0));
                if (launch_tick(r->gstream[g], r->d_desc + (size_t)i * r->n_streams + s0, sh, &tt, r->launches, r->stages)) return -1;
            }
        }
        for (u32 g = 0; g < G; g++) {
            HIP_TRY(hipEventRecord(r->gdone[g], r->gstream[g]));
            HIP_TRY(hipStreamWaitEvent(r->e->stream, r->gdone[g], 0));
        }
    }
    HIP_TRY(hipEventRecord(r->ev_end, r->e->stream));
    return 0;
}

int h264bsdmiReplaySetGroups(h264bsdmi_replay *r, u32 n_groups)
{
    if (!r || n_groups < 1 || n_groups > 8) return -1;
    std::lock_guard<std::mutex> lk(r->e->mu);
    HIP_TRY(hipSetDevice(r->e->device));
    while (r->timers.size() < (size_t)n_groups * r->n_pics) {
        TickTimers t;
        for (auto &ev : t.ev) HIP_TRY(hipEventCreate(&ev));
        r->timers.push_back(t);
    }
    for (u32 g = 0; g < n_groups; g++) {
        if (!r->gstream[g] && !r->take_stream(&r->gstream[g], false, 0)) return -1;
        if (!r->gdone[g]) HIP_TRY(hipEventCreateWithFlags(&r->gdone[g], hipEventDisableTiming));
    }
    r->n_groups = n_groups;
    return 0;
}

int h264bsdmiReplaySync(h264bsdmi_replay *r)
{
    if (!r) return -1;
    HIP_TRY(hipSetDevice(r->e->device));
    HIP_TRY(hipStreamSynchronize(r->e->stream));
    return 0;
}

int h264bsdmiReplayTimings(h264bsdmi_replay *r, float out_ms[6], u32 launches[5])
{
    if (!r) return -1;
    HIP_TRY(hipSetDevice(r->e->device));
    HIP_TRY(hipStreamSynchronize(r->e->stream));
    for (int k = 0; k < 6; k++) out_ms[k] = 0.f;
    for (u32 g = 0; g < r->n_groups; g++)
        for (u32 i0 = r->timed_first; i0 < r->timed_first + r->timed_count; i0++) {
            const size_t i = (size_t)g * r->n_pics + i0;
            for (int k = 0; k < 5; k++) {
                float ms;
                if (!((r->timed_mask >> k) & 1u)) continue;
                HIP_TRY(hipEventElapsedTime(&ms, r->timers[i].ev[k], r->timers[i].ev[k + 1]));
                out_ms[k] += ms;
            }
            if ((r->timed_mask & 4u) && r->overlap_dbk && !(r->stages & 8u) && r->n_groups == 1 && r->timers[i].sev[0] &&
                hipEventQuery(r->timers[i].sev[2]) == hipSuccess) {
                float ms;                                /* k_dbk ran on the side stream, next to the kernels above */
                if (hipEventElapsedTime(&ms, r->timers[i].sev[1], r->timers[i].sev[2]) == hipSuccess) out_ms[2] += ms;
            }
            if (r->timers[i].copy_timed && hipEventQuery(r->timers[i].sev[4]) == hipSuccess) {
                float ms;                                /* and so did k_copy, on a stream of its own (zero when the tick had no copy to launch) */
                if (hipEventElapsedTime(&ms, r->timers[i].sev[3], r->timers[i].sev[4]) == hipSuccess) out_ms[0] += ms;
            }
        }
    if (r->timed_count || !r->sched.empty()) HIP_TRY(hipEventElapsedTime(&out_ms[5], r->ev_begin, r->ev_end));
    if (launches) for (int k = 0; k < 5; k++) launches[k] = r->launches[k];
    return 0;
}

int h264bsdmiReplayFetch(h264bsdmi_replay *r, u32 stream, u32 slot, u8 *dst)
{
    if (!r || stream >= r->n_streams || slot >= r->n_slots) return -1;
    HIP_TRY(hipSetDevice(r->e->device));
    std::lock_guard<std::mutex> lk(r->e->mu);
    if (!r->d_planar) HIP_TRY(hipMalloc((void **)&r->d_planar, r->frame_bytes));
    hipLaunchKernelGGL(h264k::k_detile, dim3(512, 1), dim3(256), 0, r->e->stream, r->d_frames + ((size_t)stream * r->n_slots + slot) * r->frame_bytes,
                       r->d_planar, r->wmb, r->hmb, (size_t)0, (size_t)0);
    HIP_TRY(hipMemcpyAsync(dst, r->d_planar, r->frame_bytes, hipMemcpyDeviceToHost, r->e->stream));
    HIP_TRY(hipStreamSynchronize(r->e->stream));
    return 0;
}

int h264bsdmiReplayChecksums(h264bsdmi_replay *r, u32 slot, unsigned long long *sums)
{
    if (!r || slot >= r->n_slots) return -1;
    std::lock_guard<std::mutex> lk(r->e->mu);
    HIP_TRY(hipSetDevice(r->e->device));
    hipLaunchKernelGGL(h264k::k_checksum, dim3(r->n_streams), dim3(256), 0, r->e->stream,
                       r->d_frames + (size_t)slot * r->frame_bytes, (size_t)r->n_slots * r->frame_bytes, r->wmb, r->hmb, r->d_sums);
    HIP_TRY(hipMemcpyAsync(sums, r->d_sums, sizeof(unsigned long long) * r->n_streams, hipMemcpyDeviceToHost, r->e->stream));
    if (poll_errors(r->e)) return -1;
    return 0;
}

int h264bsdmiReplayConvert(h264bsdmi_replay *r, u32 slot, int fmt)
{
    if (!r || slot >= r->n_slots || fmt < 0 || fmt > 2) return -1;
    std::lock_guard<std::mutex> lk(r->e->mu);
    HIP_TRY(hipSetDevice(r->e->device));
    const uint32_t w = r->wmb * 16, h = r->hmb * 16;
    if (!r->d_conv) HIP_TRY(hipMalloc((void **)&r->d_conv, (size_t)w * h * 4 * r->n_streams));
    hipLaunchKernelGGL(h264k::k_convert_tiles, CONVERT_GRID(r->n_streams), dim3(256), 0, r->e->stream,
                       r->d_frames + (size_t)slot * r->frame_bytes, r->d_conv, r->wmb, r->hmb, fmt,
                       (size_t)r->n_slots * r->frame_bytes, (size_t)w * h);
    HIP_TRY(hipGetLastError());
    return 0;
}

int h264bsdmiReplayFetchConverted(h264bsdmi_replay *r, u32 stream, u32 *dst)
{
    if (!r || stream >= r->n_streams || !r->d_conv) return -1;
    HIP_TRY(hipSetDevice(r->e->device));
    HIP_TRY(hipStreamSynchronize(r->e->stream));
    const size_t n = (size_t)r->wmb * 16 * r->hmb * 16;
    HIP_TRY(hipMemcpy(dst, r->d_conv + (size_t)stream * n, n * 4, hipMemcpyDeviceToHost));
    return 0;
}

/* fmt 0..2: every h264bsdmiReplayRun() tick (lock-step sets, one group) is followed by the colour conversion of the
 * pictures it produced, inside the timed region; fmt < 0: off.  h264bsdmiReplayConvertTimings: k_convert time of the last run. */
int h264bsdmiReplaySetConvert(h264bsdmi_replay *r, int fmt_and_flags)
{
    /* flags (tests and A/B runs): 0x100 = no conversion launch behind the LAST tick of a run (what the conversion buffer then holds
     * is the work of the last tick's hosts), 0x200 = no hosting (every tick followed by its own conversion launch) */
    const int fmt = fmt_and_flags < 0 ? -1 : (fmt_and_flags & 0xFF);
    const bool no_trailing = fmt_and_flags >= 0 && (fmt_and_flags & 0x100), no_hosting = fmt_and_flags >= 0 && (fmt_and_flags & 0x200);
    if (!r || fmt > 2 || !r->sched.empty()) return -1;
    std::lock_guard<std::mutex> lk(r->e->mu);
    HIP_TRY(hipSetDevice(r->e->device));
    if (fmt >= 0) {
        const size_t n = (size_t)r->wmb * 16 * r->hmb * 16;
        if (!r->d_conv) HIP_TRY(hipMalloc((void **)&r->d_conv, n * 4 * r->n_streams));
        while (r->cev.size() < 2 * (size_t)r->n_pics) { hipEvent_t ev; HIP_TRY(hipEventCreate(&ev)); r->cev.push_back(ev); }
        /* Hosting.  Tick i can
         * convert the pictures of tick i - 1 while it decodes its own if no stream decodes INTO the frame buffer its previous
         * picture lies in (an IDR picture may); the stand-alone launch converts one frame buffer number for all streams, so the
         * streams have to be in step. */
        HIP_TRY(hipStreamSynchronize(r->e->stream));
        r->host_convert = !no_hosting;
        r->convert_trailing = !no_trailing;
        r->conv_waves = ((uint32_t)fmt_and_flags >> 16) & 15u;
        bool in_step = true;
        for (u32 s = 1; s < r->n_streams; s++) if (r->offsets[s] != r->offsets[0]) in_step = false;
        r->hosted.assign(r->n_pics, 0);
        for (u32 i = 1; in_step && i < r->n_pics; i++) r->hosted[i] = r->cur_slot[i] != r->cur_slot[i - 1];
        const size_t n_desc = (size_t)r->n_pics * r->n_streams;
        if (!r->d_desc_conv) HIP_TRY(hipMalloc((void **)&r->d_desc_conv, sizeof(FrameDesc) * n_desc));
        std::vector<FrameDesc> descs(n_desc);
        HIP_TRY(hipMemcpy(descs.data(), r->d_desc, sizeof(FrameDesc) * n_desc, hipMemcpyDeviceToHost));
        for (u32 i = 1; i < r->n_pics; i++)
            for (u32 s = 0; s < r->n_streams && r->hosted[i]; s++) {
                FrameDesc &d = descs[(size_t)i * r->n_streams + s];
                d.conv_src = r->d_frames + (size_t)s * r->frames_per_stream + (size_t)r->cur_slot[i - 1] * r->frame_bytes;
                d.conv_dst = r->d_conv + (size_t)s * n;
                d.conv_fmt = (uint32_t)fmt;
            }
        HIP_TRY(hipMemcpy(r->d_desc_conv, descs.data(), sizeof(FrameDesc) * n_desc, hipMemcpyHostToDevice));
    }
    r->convert_fmt = fmt;
    return 0;
}

int h264bsdmiReplayConvertTimings(h264bsdmi_replay *r, float *ms, u32 *launches)
{
    if (!r || r->convert_fmt < 0) return -1;
    HIP_TRY(hipSetDevice(r->e->device));
    HIP_TRY(hipStreamSynchronize(r->e->stream));
    *ms = 0.f; *launches = 0;
    for (u32 i = r->timed_first; i < r->timed_first + r->timed_count; i++) {
        float t;
        if (i >= r->cev_on.size() || !r->cev_on[i]) continue;      /* converted by the next tick's k_frame_dbk: no launch of its own */
        HIP_TRY(hipEventElapsedTime(&t, r->cev[2 * i], r->cev[2 * i + 1]));
        *ms += t; (*launches)++;
    }
    return 0;
}

int h264bsdmiReplaySetTimedKernels(h264bsdmi_replay *r, unsigned mask)
{
    if (!r) return -1;
    r->timed_mask = mask & 31u;     /* bit k: HIP events around kernel k (k_copy, k_recon_inter, k_dbk, k_frame_intra, k_frame_dbk) */
    return 0;
}

int h264bsdmiReplaySetStages(h264bsdmi_replay *r, unsigned mask)
{
    if (!r) return -1;
    r->stages = mask & 15u;        /* bit 3: keep k_dbk on the main stream (no overlap) */
    return 0;
}

/* Debug hook: cycle accounting of k_frame_tail's deblocking loop (workgroup 0 of the next launches).
 * out[16][8]: per wave {pick, filter, extra rounds, own-memory wait, filtered count, barrier wait}. */
int h264bsdmiDebugTailProfile(int enable, unsigned long long *out)
{
    Engine *e = engine_get();
    if (!e) return -1;
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipDeviceSynchronize());
    if (enable) {
        if (!e->tail_prof) HIP_TRY(hipMalloc((void **)&e->tail_prof, (16 * 16 + 16 * 8) * sizeof(unsigned long long)));
        HIP_TRY(hipMemset(e->tail_prof, 0, (16 * 16 + 16 * 8) * sizeof(unsigned long long)));
        HIP_TRY(hipDeviceSynchronize());
    } else if (e->tail_prof) {
        if (out) HIP_TRY(hipMemcpy(out, e->tail_prof, (16 * 16 + 16 * 8) * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        HIP_TRY(hipFree(e->tail_prof));
        e->tail_prof = nullptr;
    }
    return 0;
}

/* Test / tuning hook: how the per-picture kernels split pictures from now on (TailConfig; descriptors built earlier keep their
 * bands): rows per band for light and heavy pictures (0 = one band) and wavefronts per workgroup, for k_frame_dbk and
 * k_frame_intra.  A value of 0xFFFFFFFF leaves that setting alone. */
int h264bsdmiDebugSetTail(u32 dbk_rows_light, u32 dbk_rows_heavy, u32 dbk_waves, u32 intra_rows_light, u32 intra_rows_heavy, u32 intra_waves, u32 band_budget)
{
    (void)tail_config();                                     /* the environment first, once */
    std::lock_guard<std::mutex> lk(g_tail_mu);
    if (dbk_rows_light != 0xFFFFFFFFu) g_tail.dbk_rows_light = dbk_rows_light;
    if (dbk_rows_heavy != 0xFFFFFFFFu) g_tail.dbk_rows_heavy = dbk_rows_heavy;
    if (dbk_waves != 0xFFFFFFFFu && dbk_waves >= 1) g_tail.dbk_waves = dbk_waves;
    if (intra_rows_light != 0xFFFFFFFFu) g_tail.intra_rows_light = intra_rows_light;
    if (intra_rows_heavy != 0xFFFFFFFFu) g_tail.intra_rows_heavy = intra_rows_heavy;
    if (intra_waves != 0xFFFFFFFFu && intra_waves >= 1) g_tail.intra_waves = intra_waves;
    if (band_budget != 0xFFFFFFFFu) g_tail.band_budget = band_budget;
    return 0;
}

unsigned long long h264bsdmiReplayJobBytes(h264bsdmi_replay *r) { return r ? r->job_bytes : 0; }
u32 h264bsdmiReplayFrameBytes(h264bsdmi_replay *r) { return r ? r->frame_bytes : 0; }

} /* extern "C" */
