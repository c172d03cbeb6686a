/*
 * kernels.hip.h — hand-written gfx950 (CDNA4, wave64) kernels of the macroblock reconstruction path.
 *
 * Frames live in HBM as MACROBLOCK TILES: 384 contiguous bytes per macroblock (Y 16x16 | Cb 8x8 | Cr 8x8), tiles in
 * macroblock address order — see "frame layout in HBM" below.  Planar I420 exists only where pictures leave the device.
 *
 *   k_copy        : inter macroblocks that are whole-sample copies without residual (host-built list of runs of up to
 *                   8 tiles; 61 % of the macroblocks of the 1080p stream): a zero-displacement run is one contiguous
 *                   block of count x 384 bytes, 16 bytes per lane per access, every load before the first store.
 *                       reference: the P_Skip / integer-mv path of src/h264bsd_inter_prediction.c:361-482,
 *                                  h264bsdFillBlock src/h264bsd_reconstruct.c:2244, src/h264bsd_image.c:81
 *   k_recon_inter : every other inter macroblock, one wavefront each (= one workgroup: the wavefronts share nothing), list-driven.
 *                   Reference windows staged in LDS tile row by tile row (aligned 16-byte loads), 6-tap luma on packed
 *                   sample pairs / bilinear chroma, dequant + 4x4 inverse transform with DPP quad transposes, residual
 *                   add; the macroblock leaves as its tile (24 x 16 bytes).
 *                       reference: src/h264bsd_reconstruct.c, src/h264bsd_inter_prediction.c:361-482,
 *                                  src/h264bsd_transform.c, src/h264bsd_image.c:172
 *   k_dbk         : boundary strengths + alpha / beta / tc0 VALUES of every macroblock the host could not prove strength-free
 *                   (metadata only, four MBs per wavefront) -> 48-byte deblocking records + one flag byte per MB
 *                   (DBKF_*: filtered at all / touches the left / the upper neighbour / has an active inner edge).
 *                       reference: src/h264bsd_deblocking.c:1187-1541
 *   k_frame_intra : ONE workgroup per picture (a picture never leaves its CU): intra and concealed macroblocks,
 *                   dataflow-scheduled in LDS (dependency counters + ready queue, a free wavefront takes one ready MB).
 *                       reference: src/h264bsd_intra_prediction.c, src/h264bsd_conceal.c
 *   k_frame_dbk   : ONE workgroup per picture: the in-loop filter, dataflow-scheduled in LDS at macroblock-EDGE
 *                   granularity (a macroblock waits only for the neighbours whose samples it really shares), eight-lane
 *                   workers (up to eight macroblocks per wavefront step), two ready lists, packed 16-bit arithmetic.
 *                       reference: src/h264bsd_deblocking.c:575-1745
 *   k_convert / k_output / k_detile : YUV420 -> RGBA / BGRA / YCbCrA, cropped windows, tiles -> planar I420.
 *                       reference: src/h264bsd_decoder.c:1163-1370, :970-1001
 *   k_checksum    : position-weighted 64-bit checksum of a frame in planar order (on-device verification).
 *
 * Everything is integer arithmetic on u8 samples / i16 levels / i32 intermediates.  The one contraction on the path — the 6-tap
 * filter, a Toeplitz product — is a fifth of k_recon_inter's vector instructions, and an i8 MFMA would take over half of THAT
 * only for the one-dimensional classes ("the matrix pipe"): no MFMA.
 */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "framejob.h"

#include "kernels/common.hip.h"
#include "kernels/k_dbk.hip.h"
#include "kernels/k_copy.hip.h"
#include "kernels/k_recon_inter.hip.h"
#include "kernels/convert.hip.h"
#include "kernels/tail_common.hip.h"
#include "kernels/k_frame_intra.hip.h"
#include "kernels/k_frame_dbk.hip.h"
#include "kernels/k_pixels_io.hip.h"
