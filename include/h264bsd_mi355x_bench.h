/*
 * h264bsd_mi355x_bench.h — measurement and test harness entry points.  NOT part of the product ABI: they are exported
 * by libh264bsd_mi355x_bench.so only (the same objects as libh264bsd_mi355x.so linked with a wider export list), which
 * bench.py, the kernel-level tests and the Python mirror load.  Plain C ABI.
 */
#ifndef H264BSD_MI355X_BENCH_H
#define H264BSD_MI355X_BENCH_H

#include "h264bsd_mi355x.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Complete a frame job built outside the parser (tests, tools): given a buffer whose FjHeader geometry /
 * rec_off / mv_off / coef_off, records, motion vectors and n_coef_blocks coefficient blocks are filled in,
 * derive the schedules (intra levels, copy runs, general-inter list, deblocking index), the compact form of the
 * motion vectors (one vector in the record of a macroblock that has one, a sparse section for the others: framejob.h)
 * and total_bytes exactly as the parser does.  The dense array int16 mv[n_mbs][16][2] at mv_off is INPUT only: it
 * must lie in front of coef_off or behind everything else the job can grow to (capacity: + 64 bytes per macroblock
 * for the sparse section); nothing on the device reads it.  cur_slot / n_slots / is_idr stay as the caller set them.
 * 0 = ok. */
int h264bsdmiJobFinalize(u8 *job, u32 capacity, u32 n_coef_blocks);

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
/* Give an existing set another schedule (other first pictures, heavy lanes, stream groups; arguments as above) without
 * allocating and uploading its jobs again: the measurement legs of bench.py share one set.  Frame buffers and deblocking
 * scratch are zeroed like those of a new set; whatever the old schedule had enqueued is waited for.  0 = ok. */
int  h264bsdmiReplayReschedule(h264bsdmi_replay *r, const u32 *offsets, u32 heavy_lanes, u32 heavy_delay, u32 groups);
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
/* BASELINE.json config 3 ("ARGB conversion on-GPU"): fmt 0 RGBA, 1 BGRA (= the ARGB word), 2 YCbCrA: every picture a tick of
 * h264bsdmiReplayRun() produces is converted inside the timed region (1024 B written per macroblock) — by wavefronts of the NEXT
 * tick's k_frame_dbk workgroups, beside that picture's filtering, where the schedule allows it (streams in step, the next picture
 * is not decoded into the same frame buffer), by a k_convert_tiles launch behind its own tick otherwise (always for the last tick
 * of a run); fmt < 0 switches it off.  Flags or-ed into fmt (tests, A/B): 0x100 = no launch behind the last tick of a run,
 * 0x200 = no hosting, bits 16..19 = conversion wavefronts per k_frame_dbk workgroup (0 = the default).  ConvertTimings: HIP-event time and number of the k_convert_tiles launches of the last run. */
int  h264bsdmiReplaySetConvert(h264bsdmi_replay *r, int fmt);
int  h264bsdmiReplayConvertTimings(h264bsdmi_replay *r, float *ms, u32 *launches);
/* Debug hook: cycle accounting of the per-picture kernels (k_frame_intra, k_frame_dbk) for workgroup 0 of every launch between
 * enable=1 and enable=0 (which copies out[16 waves][8]: cycles in {choose MB, filter, extra rounds, wait for
 * own memory traffic, #filtered, barrier wait}). */
int  h264bsdmiDebugTailProfile(int enable, unsigned long long *out);
/* Test / tuning hook: how the two per-picture kernels split the pictures of replay sets and decoder instances created
 * from now on into row bands (one workgroup per band): rows per band for light and for heavy pictures (more than a quarter of
 * the macroblocks intra coded; 0 = one band per picture) and wavefronts per workgroup, for k_frame_dbk and for k_frame_intra;
 * band_budget: most workgroups of one launch (bands per picture <= band_budget / pictures of the tick, at least 1).
 * 0xFFFFFFFF leaves a setting alone.  Also settable through H264BSDMI_TAIL="a,b,c,d,e,f" / H264BSDMI_BAND_BUDGET. */
int  h264bsdmiDebugSetTail(u32 dbk_rows_light, u32 dbk_rows_heavy, u32 dbk_waves, u32 intra_rows_light, u32 intra_rows_heavy, u32 intra_waves,
                           u32 band_budget);
/* How often a tripwire of the kernels (DEVERR_*, h264bsdmiDeviceErrors) has fired on all devices since the library was loaded,
 * after waiting for the devices.  Monotonic — the bits of h264bsdmiDeviceErrors() are sticky and cannot show that a bit which a
 * hand-built test job set earlier fired again: tests assert that their own pictures add no event.  0xFFFFFFFF = HIP error. */
unsigned h264bsdmiDebugDeviceErrorEvents(void);
/* Bytes of packed syntax (frame jobs) per stream and of one frame, for the byte accounting. */
unsigned long long h264bsdmiReplayJobBytes(h264bsdmi_replay *r);
u32  h264bsdmiReplayFrameBytes(h264bsdmi_replay *r);
#ifdef __cplusplus
}
#endif
#endif
