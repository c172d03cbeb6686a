/*
 * h264bsd_mi355x.h — extensions of the drop-in API that only exist because the pixel path runs on
 * an MI355X: frame-job capture, explicit batching and the HBM-resident replay used for throughput
 * measurement.  Plain C ABI: pointers and sizes only.
 *
 * None of these has a counterpart in the reference (it has no device, no batching: SURVEY.md §2);
 * the seams they expose are the reference's internal ones:
 *   frame job  = input of h264bsdDecodeMacroblock (src/h264bsd_macroblock_layer.c:965) for every
 *                macroblock of a picture + input of h264bsdFilterPicture (src/h264bsd_deblocking.c:575)
 */
#ifndef H264BSD_MI355X_EXT_H
#define H264BSD_MI355X_EXT_H

#include "h264bsd_decoder.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- capture: run only the host parser, hand every finished picture's packed frame job to cb ----
 * No GPU is touched.  h264bsdDecode()'s return codes are unchanged; h264bsdNextOutputPicture*()
 * return NULL (there are no pixels).  cb's blob pointer is valid only during the call. */
typedef void (*h264bsdmi_job_cb)(void *user, const u8 *blob, u32 bytes);
u32 h264bsdmiInitCapture(storage_t *pStorage, u32 noOutputReordering, h264bsdmi_job_cb cb, void *user);
/* Pop the next picture of the output queue like h264bsdNextOutputPicture() (reference
 * src/h264bsd_decoder.c:1045-1066 -> h264bsdDpbOutputPicture, src/h264bsd_dpb.c:1415), but return the DPB
 * slot that holds it (the FjHeader.cur_slot of the job that wrote it) instead of pixels; -1 when the queue
 * is empty.  Works in capture mode, where it is the only way to observe the output order. */
int h264bsdmiNextOutputInfo(storage_t *pStorage, u32 *picId, u32 *isIdrPic, u32 *numErrMbs);

/* Complete a frame job built outside the parser (tests, tools): given a buffer whose FjHeader geometry /
 * rec_off / mv_off / coef_off, records, motion vectors and n_coef_blocks coefficient blocks are filled in,
 * derive the schedules (intra levels, copy runs, general-inter list, deblocking index) and total_bytes exactly
 * as the parser does.  cur_slot / n_slots / is_idr stay as the caller set them.  0 = ok. */
int h264bsdmiJobFinalize(u8 *job, u32 capacity, u32 n_coef_blocks);

/* ---- device-resident output ----
 * The reference hands pictures over as host pointers (h264bsdNextOutputPicture*, src/h264bsd_decoder.c:1045-1161)
 * and leaves cropping to the application (h264bsdCroppingParams, :970-1001).  On an MI355X box the consumer of
 * decoded video is normally another GPU program, so this variant pops the next output picture like
 * h264bsdNextOutputPicture() but leaves it in HBM: 3.13 MB per 1080p frame never cross PCIe. */
#define H264BSDMI_FMT_RGBA   0   /* bytes R,G,B,A   (h264bsdConvertToRGBA)   */
#define H264BSDMI_FMT_BGRA   1
#define H264BSDMI_FMT_YCBCRA 2
#define H264BSDMI_FMT_I420   3   /* planar Y, Cb, Cr as the reference's u8* picture */
typedef struct h264bsdmi_device_picture {
    void *data;               /* DEVICE pointer; valid until the next h264bsdDecode()/h264bsdShutdown() of this instance */
    u32   width, height;      /* in samples, after cropping when requested                               */
    u32   pitch;              /* bytes per row: width for I420 luma (chroma planes: width/2), 4*width otherwise */
    u32   format;
    u32   picId, isIdrPic, numErrMbs;
    void *stream;             /* hipStream_t the producing work ran on; it has been synchronised on return */
} h264bsdmi_device_picture;
/* format: H264BSDMI_FMT_*; crop != 0 applies the SPS frame-cropping rectangle on the device.
 * I420 without cropping is zero-copy (the pointer aims into the decoded-picture buffer); every other combination
 * is produced by one kernel into a per-instance HBM buffer.  Returns 1 = picture, 0 = no picture, <0 = error. */
int h264bsdmiNextOutputPictureDevice(storage_t *pStorage, int format, int crop, h264bsdmi_device_picture *out);

/* ---- host parse pipeline at scale ----
 * h264bsdDecode() consumes one NAL unit of one stream per call; a caller that feeds hundreds of streams needs the
 * loop of posix/test_h264bsd.c:146-177 for each of them and its own threading.  These entry points move both into
 * the library. */
/* Feed NAL units of one instance until a picture is complete: calls h264bsdDecode() on buf, buf+readBytes, ...
 * and stops after H264BSD_PIC_RDY or when the buffer is used up.  H264BSD_RDY / H264BSD_HDRS_RDY continue, errors
 * are counted in *nErrors (may be NULL) and skipped like the reference harness does.  Returns the last status
 * (H264BSD_PIC_RDY, or H264BSD_RDY at the end of the buffer); *consumed = bytes to advance. */
u32 h264bsdmiDecodePicture(storage_t *pStorage, u8 *buf, u32 len, u32 picId, u32 *consumed, u32 *nErrors);
/* The same for n independent instances at once, on the library's parser threads (the caller's thread helps).
 * status[i] / consumed[i] / nErrors[i] (nErrors may be NULL) as above.  Instances must be distinct. */
int h264bsdmiDecodePictureBatch(u32 n, storage_t *const *pStorage, u8 *const *buf, const u32 *len, const u32 *picId,
                                u32 *status, u32 *consumed, u32 *nErrors);
/* Number of parser threads (default: online CPUs, at most 64; env H264BSDMI_THREADS).  Returns the value in use. */
int h264bsdmiSetParserThreads(int n);

/* ---- device engine ---- */
/* Number of usable GPUs (0 when the HIP runtime finds none); selects the device for this process. */
int  h264bsdmiDeviceCount(void);
int  h264bsdmiSetDevice(int device);
/* Run every queued frame job of every decoder instance of this process now (they are otherwise run
 * lazily, when the first picture is pulled).  Returns 0 on success. */
int  h264bsdmiFlush(void);
/* Sticky device error bits (0 = none): 1 a residual outside [-512,511] reached the kernels (the host parser finds this
 * decode error while it parses, so the kernels' check is a tripwire), 2 / 4 the intra / deblocking scheduler of a
 * picture gave up.  Any bit means that pixels were produced that cannot be trusted; the library also says so on stderr. */
unsigned h264bsdmiDeviceErrors(void);
/* Like h264bsdmiFlush() but returns as soon as the copies and kernels are enqueued, so that parsing the next
 * pictures overlaps the reconstruction of these.  Any call that needs pixels (h264bsdNextOutputPicture*,
 * h264bsdmiFlush) waits for the outstanding work first. */
int  h264bsdmiFlushAsync(void);

/* ---- HBM-resident replay (bench / parity tests): kernels only, no host parsing in the loop ----
 * A replay set holds n_streams independent copies of one captured stream: every copy owns private
 * frame jobs and a private DPB in HBM.  One "tick" reconstructs + deblocks picture k of all streams
 * in batched launches. */
typedef struct h264bsdmi_replay h264bsdmi_replay;
/* blobs[i]/bytes[i]: the n_pics frame jobs of ONE stream in decode order (from h264bsdmiInitCapture). */
h264bsdmi_replay *h264bsdmiReplayCreate(const u8 *const *blobs, const u32 *bytes, u32 n_pics, u32 n_streams);
/* The "staggered" variant of the many-streams workload: odd-numbered streams run picture
 * (k + odd_offset) mod n_pics in tick k, so a tick mixes two different pictures (e.g. an IDR picture of one
 * half of the streams with a P picture of the other half).  odd_offset must index an IDR picture. */
h264bsdmi_replay *h264bsdmiReplayCreateStaggered(const u8 *const *blobs, const u32 *bytes, u32 n_pics, u32 n_streams,
                                                 u32 odd_offset);
/* Streams that are not in step at all: stream s starts at picture offsets[s] (NULL = 0 for all) and wraps around
 * (picture 0 must be an IDR picture).  heavy_lanes = 0: tick k holds picture (k + offsets[s]) mod n_pics of every
 * stream, so every tick lasts as long as its slowest picture.  heavy_lanes = 1..8: mostly intra-coded pictures
 * (more than a quarter of their macroblocks) leave the common tick and run on one of heavy_lanes extra HIP streams;
 * their stream rejoins the common ticks heavy_delay ticks later (guarded by an event).  h264bsdmiReplayRun() then
 * always runs one whole lap (every stream n_pics pictures) and h264bsdmiReplayTimings() reports only the total. */
h264bsdmi_replay *h264bsdmiReplayCreateDesync(const u8 *const *blobs, const u32 *bytes, u32 n_pics, u32 n_streams,
                                              const u32 *offsets, u32 heavy_lanes, u32 heavy_delay);
/* The same with the streams split into `groups` groups (stream s -> group s % groups) that run their own ticks on their
 * own HIP streams (plus the heavy lanes): a group's tick lasts as long as its own slowest picture and the other groups'
 * workgroups fill the compute units it leaves idle.  groups == 1 is h264bsdmiReplayCreateDesync. */
h264bsdmi_replay *h264bsdmiReplayCreateSched(const u8 *const *blobs, const u32 *bytes, u32 n_pics, u32 n_streams,
                                             const u32 *offsets, u32 heavy_lanes, u32 heavy_delay, u32 groups);
void h264bsdmiReplayDestroy(h264bsdmi_replay *r);
/* Enqueue ticks [first, first+count) on the engine stream; asynchronous.  0 = ok. */
int  h264bsdmiReplayRun(h264bsdmi_replay *r, u32 first, u32 count);
/* Wait for everything enqueued. */
int  h264bsdmiReplaySync(h264bsdmi_replay *r);
/* Copy the current content of DPB slot `slot` of stream `stream` to host memory (frame_bytes). */
int  h264bsdmiReplayFetch(h264bsdmi_replay *r, u32 stream, u32 slot, u8 *dst);
/* 64-bit checksum (computed on the device) of slot `slot` of every stream into sums[n_streams]. */
int  h264bsdmiReplayChecksums(h264bsdmi_replay *r, u32 slot, unsigned long long *sums);
/* On-device colour conversion of slot `slot` of every stream (fmt 0 RGBA, 1 BGRA, 2 YCbCrA) into the
 * set's ARGB planes; fetch one with ...FetchConverted. */
int  h264bsdmiReplayConvert(h264bsdmi_replay *r, u32 slot, int fmt);
int  h264bsdmiReplayFetchConverted(h264bsdmi_replay *r, u32 stream, u32 *dst);
/* HIP-event timing (events on the engine's own stream) of the last h264bsdmiReplayRun(), summed over its
 * ticks: out_ms[0..4] = k_copy, k_recon_inter, k_dbk, k_frame_intra, k_frame_dbk; out_ms[5] = whole run;
 * launches[0..4] = number of launches of each kernel.  k_dbk runs on a second stream concurrently with the
 * reconstruction kernels, so the five times add up to more than out_ms[5]. */
int  h264bsdmiReplayTimings(h264bsdmi_replay *r, float out_ms[6], u32 launches[5]);
/* Split the streams of the set into n_groups (1..8) groups, each on its own HIP stream, so that the
 * latency-bound per-picture kernel of one group overlaps the throughput-bound kernels of another.
 * With more than one group the per-class times of h264bsdmiReplayTimings() are sums over concurrently
 * running launches (they exceed out[3], the whole run). */
int  h264bsdmiReplaySetGroups(h264bsdmi_replay *r, u32 n_groups);
/* Test hook: which stages h264bsdmiReplayRun() launches: bit0 inter reconstruction, bit1 intra
 * reconstruction, bit2 deblocking (default 7 = all); bit3: keep k_dbk on the main stream instead of overlapping
 * it with the reconstruction kernels on a second stream (A/B measurements). */
int  h264bsdmiReplaySetStages(h264bsdmi_replay *r, unsigned mask);
/* Which kernels of a tick are bracketed by HIP events (bit k = kernel k in the order of h264bsdmiReplayTimings,
 * default 31 = all).  Every event is a barrier packet between two kernels; the bench times all five kernels in
 * its warm-up steps and only the dominant one in the timed steps. */
int  h264bsdmiReplaySetTimedKernels(h264bsdmi_replay *r, unsigned mask);
/* BASELINE.json config 3 ("ARGB conversion on-GPU"): fmt 0 RGBA, 1 BGRA (= the ARGB word), 2 YCbCrA: every tick of
 * h264bsdmiReplayRun() is followed, inside the timed region, by the colour conversion of the pictures it produced
 * (k_convert, 1024 B written per macroblock); fmt < 0 switches it off.  ConvertTimings: HIP-event time of those launches. */
int  h264bsdmiReplaySetConvert(h264bsdmi_replay *r, int fmt);
int  h264bsdmiReplayConvertTimings(h264bsdmi_replay *r, float *ms, u32 *launches);
/* Debug hook: cycle accounting of k_frame_tail's deblocking loop for workgroup 0 of every launch between
 * enable=1 and enable=0 (which copies out[16 waves][8]: cycles in {choose MB, filter, extra rounds, wait for
 * own memory traffic, #filtered, barrier wait}). */
int  h264bsdmiDebugTailProfile(int enable, unsigned long long *out);
/* Bytes of packed syntax (frame jobs) per stream and of one frame, for the byte accounting. */
unsigned long long h264bsdmiReplayJobBytes(h264bsdmi_replay *r);
u32  h264bsdmiReplayFrameBytes(h264bsdmi_replay *r);

#ifdef __cplusplus
}
#endif
#endif
