/*
 * h264bsd_decoder.h — public C API of the MI355X-native decoder back end.
 *
 * Drop-in for the reference's src/h264bsd_decoder.h:45-93 (same 21 exported functions, same enum
 * values, same call protocol; Windows export list win/h264bsd.def:3-18).  Each prototype below names
 * the reference definition it replaces.  Differences a caller can observe:
 *   - storage_t is opaque (same size, 4648 bytes on LP64, so it can still live on the caller's
 *     stack as in posix/test_h264bsd.c:129);
 *   - like the reference, h264bsdDecode() removes the emulation-prevention bytes of the NAL unit it extracts IN the
 *     caller's buffer (byte_stream.c:193-235): a caller that decodes a buffer twice keeps a private copy, and every
 *     decoder instance needs its own writable copy (h264bsdmiSetInputReadOnly() in h264bsd_mi355x.h switches this off);
 *   - pixels are produced on the GPU: h264bsdDecode() only parses and queues a frame job, the
 *     picture is materialised (and copied to host memory) by h264bsdNextOutputPicture*();
 *   - damaged streams are handled like the reference: H264BSD_ERROR for the broken NAL unit — including the one
 *     error the reference finds after dequantisation, a residual sample outside [-512,511]
 *     (src/h264bsd_transform.c:184-188), which the host parser decides while it parses — the same macroblocks
 *     rolled back (src/h264bsd_slice_data.c:298-354) and concealed (src/h264bsd_conceal.c), numErrMbs reported.
 *     No deviation from the reference (run with zeroed allocations) is known.
 */
#ifndef H264BSD_MI355X_DECODER_H
#define H264BSD_MI355X_DECODER_H

#ifdef __cplusplus
extern "C" {
#endif

#ifndef BASETYPE_H_INCLUDED            /* reference src/basetype.h:28-33 */
#define BASETYPE_H_INCLUDED
typedef unsigned char  u8;
typedef signed char    i8;
typedef unsigned short u16;
typedef signed short   i16;
typedef unsigned int   u32;
typedef signed int     i32;
#endif

#ifndef HANTRO_OK                      /* reference src/h264bsd_util.h:54-55 */
#define HANTRO_OK    0
#define HANTRO_NOK   1
#define HANTRO_FALSE 0
#define HANTRO_TRUE  1
#endif

/* reference src/h264bsd_decoder.h:45-52 */
enum {
    H264BSD_RDY,
    H264BSD_PIC_RDY,
    H264BSD_HDRS_RDY,
    H264BSD_ERROR,
    H264BSD_PARAM_SET_ERROR,
    H264BSD_MEMALLOC_ERROR
};

/* Opaque decoder instance; sizeof == sizeof(reference storage_t) on LP64 (4648). */
typedef struct storage {
    void *opaque;
    unsigned char reserved[4640];
} storage_t;

u32  h264bsdInit(storage_t *pStorage, u32 noOutputReordering);                  /* decoder.c:90  */
u32  h264bsdDecode(storage_t *pStorage, u8 *byteStrm, u32 len, u32 picId,
                   u32 *readBytes);                                             /* decoder.c:152 */
void h264bsdShutdown(storage_t *pStorage);                                      /* decoder.c:534 */

u8  *h264bsdNextOutputPicture(storage_t *pStorage, u32 *picId, u32 *isIdrPic,
                              u32 *numErrMbs);                                  /* decoder.c:599 */
u32 *h264bsdNextOutputPictureRGBA(storage_t *pStorage, u32 *picId, u32 *isIdrPic,
                                  u32 *numErrMbs);                              /* decoder.c:648 */
u32 *h264bsdNextOutputPictureBGRA(storage_t *pStorage, u32 *picId, u32 *isIdrPic,
                                  u32 *numErrMbs);                              /* decoder.c:690 */
u32 *h264bsdNextOutputPictureYCbCrA(storage_t *pStorage, u32 *picId, u32 *isIdrPic,
                                    u32 *numErrMbs);                            /* decoder.c:732 */

u32  h264bsdPicWidth(storage_t *pStorage);                /* in macroblocks, decoder.c:771  */
u32  h264bsdPicHeight(storage_t *pStorage);               /* in macroblocks, decoder.c:806  */
u32  h264bsdVideoRange(storage_t *pStorage);                                   /* decoder.c:893 */
u32  h264bsdMatrixCoefficients(storage_t *pStorage);                           /* decoder.c:928 */
void h264bsdCroppingParams(storage_t *pStorage, u32 *croppingFlag, u32 *left, u32 *width,
                           u32 *top, u32 *height);                             /* decoder.c:970 */
void h264bsdSampleAspectRatio(storage_t *pStorage, u32 *sarWidth, u32 *sarHeight); /* decoder.c:1019 */
u32  h264bsdCheckValidParamSets(storage_t *pStorage);                          /* decoder.c:864 */
void h264bsdFlushBuffer(storage_t *pStorage);                                  /* decoder.c:834 */
u32  h264bsdProfile(storage_t *pStorage);                                      /* decoder.c:1084 */

storage_t *h264bsdAlloc(void);                                                 /* decoder.c:1110 */
void h264bsdFree(storage_t *pStorage);                                         /* decoder.c:1133 */

/* Stateless colour conversion of one I420 frame (width/height in samples, multiples of 2):
 * runs on the GPU (data and pOutput are host pointers).  decoder.c:1163/1244/1324 */
void h264bsdConvertToRGBA(u32 width, u32 height, u8 *data, u32 *pOutput);
void h264bsdConvertToBGRA(u32 width, u32 height, u8 *data, u32 *pOutput);
void h264bsdConvertToYCbCrA(u32 width, u32 height, u8 *data, u32 *pOutput);

#ifdef __cplusplus
}
#endif
#endif
