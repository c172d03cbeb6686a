/*
 * h264bsd_mi355x.h — extensions of the drop-in API that only exist because the pixel path runs on
 * an MI355X: frame-job capture, explicit batching, device-resident output.  Plain C ABI: pointers and sizes only.
 * (The HBM-resident replay sets used for throughput measurement and kernel tests are NOT part of the product library:
 * include/h264bsd_mi355x_bench.h, libh264bsd_mi355x_bench.so.)
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
 * Frames live in HBM as macroblock tiles, so every picture that is handed out is laid out by one kernel into a
 * per-instance HBM buffer (planar I420 by k_detile, windows and conversions by k_output).  Returns 1 = picture, 0 = no picture, <0 = error. */
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
/* h264bsdNextOutputPicture() (src/h264bsd_decoder.c:599-646) of n distinct instances at once, on the same threads: every thread
 * enqueues its instances' pictures on their way out (layout kernel + copy into pinned host memory) and waits for ITS copies
 * only — the engine's lock is held while enqueueing, never while waiting — so the 3.1 MB transfers of the instances overlap.
 * pictures[i] = NULL when instance i has no picture to give; picId / isIdrPic / numErrMbs may be NULL.  0 = ok. */
int h264bsdmiNextOutputPictureBatch(u32 n, storage_t *const *pStorage, u8 **pictures, u32 *picId, u32 *isIdrPic, u32 *numErrMbs);
/* Both in one call — one round of the reference harness's per-stream loop (posix/test_h264bsd.c:146-177: take the pictures that are
 * ready, then decode on) for n distinct instances: every thread FIRST pulls its instance's next output picture (as above:
 * pictures[i], outPicId[i], outIsIdrPic[i], outNumErrMbs[i]; the last three may be NULL) and THEN parses that instance's next
 * picture (as h264bsdmiDecodePicture: buf / len / picId -> status / consumed / nErrors).  The pictures of some instances cross the
 * link while other instances are parsed; with the two calls above the CPUs wait for the link and the link for the CPUs.
 * An instance that still has FURTHER pictures waiting in its output queue is not fed — the next slice would discard them, as in the
 * reference (src/h264bsd_dpb.c:1260-1261): status[i] = H264BSD_RDY, consumed[i] = 0, call again.  len[i] = 0: pull only.
 * pictures[i] stays valid until the next pull from instance i (it is the pinned host mirror of the picture's DPB slot, which the
 * parsing of later pictures does not touch, and which survives the activation of a new sequence parameter set) — longer than the
 * reference's "until the next h264bsdDecode()"; its dimensions are those the instance reported BEFORE the call.  0 = ok. */
int h264bsdmiPullAndDecodePictureBatch(u32 n, storage_t *const *pStorage, u8 **pictures, u32 *outPicId, u32 *outIsIdrPic, u32 *outNumErrMbs,
                                       u8 *const *buf, const u32 *len, const u32 *picId, u32 *status, u32 *consumed, u32 *nErrors);
/* Number of parser threads (default: the CPUs the process may use — affinity mask, cgroup quota + a quarter — divided by H264BSDMI_HOST_SHARE,
 * at most 64; env H264BSDMI_THREADS).  Returns the value in use.  With the default, batches that PULL pictures run on at most as many of
 * these threads as the process may have running at once (the quota itself); a count named here or in the environment is used as it is. */
int h264bsdmiSetParserThreads(int n);
/* INPUT BUFFERS ARE MODIFIED by h264bsdDecode() and by the two calls above, exactly as by the reference: the emulation-
 * prevention bytes of the NAL unit just parsed are removed IN the caller's buffer (src/h264bsd_byte_stream.c), so that a
 * caller who feeds the same bytes again sees what the reference's caller sees.  Consequences: every decoder instance needs
 * its own private, writable copy of the stream (never hand one buffer to several instances of a batch, never a read-only
 * mapping).  on != 0 switches the write-back off for this instance: the buffer may then be shared and read-only; pictures are
 * the same, the h264bsdDecode() call trace can differ from the reference's only where a NAL unit that contains emulation-
 * prevention bytes is fed a second time.  Returns 0. */
int h264bsdmiSetInputReadOnly(storage_t *pStorage, u32 on);
/* Copy elision.  The parser knows what every macroblock tile of every frame buffer holds: a P macroblock that copies the
 * co-located macroblock of its reference (zero motion, no residual) and that no deblocking edge touches leaves its tile
 * equal to the reference's, and frame buffers are reused in rotation — so the buffer a picture is decoded into often holds
 * exactly those bytes already (a region that has not changed since the buffer's previous picture).  Such copies are left out
 * of the frame job: nothing is read, nothing is written, the picture is bit-identical.  ON by default for decoders bound
 * to a device (env H264BSDMI_COPY_ELISION=0 switches it off for the process), OFF by default in capture mode, where a
 * frame job is then a pure function of its picture; a capture that is replayed IN ORDER from an IDR picture onto frames
 * that persist (the bench harness's replay sets) may switch it on.  Call before the first h264bsdDecode().  Returns 0. */
int h264bsdmiSetCopyElision(storage_t *pStorage, u32 on);

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

#ifdef __cplusplus
}
#endif
#endif
