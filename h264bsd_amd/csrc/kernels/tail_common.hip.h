/* kernels/tail_common.hip.h — what the two per-picture kernels share: wavefront counts, tickets, row bands.  Part of kernels.hip.h (which see); not a stand-alone header. */
#pragma once
namespace h264k {
/* ONE workgroup per picture in the two kernels below (a picture never leaves its CU): dependencies inside a picture
 * are tracked in LDS, no kernel boundary and no inter-workgroup traffic inside a picture.  Occupancy comes from
 * batching streams (256 pictures = one workgroup per CU).  k_frame_intra: 12 wavefronts = 162 VGPRs without spills
 * (16 wavefronts cap the kernel at 128 VGPRs: 33 spilled, 20.8 vs 19.3 ms per step; 8: 22.2 ms). */
#ifndef TAIL_WAVES_N
#define TAIL_WAVES_N 12
#endif
constexpr int TAIL_WAVES = TAIL_WAVES_N;
#ifndef DBK_WAVES_N
#define DBK_WAVES_N 12
#endif
constexpr int DBK_WAVES = DBK_WAVES_N;              /* most wavefronts of k_frame_dbk (the launch chooses: TailConfig.dbk_waves); a wavefront
                                                        is eight workers (deblock_mb) */

/* Which picture and which row band a workgroup of the two per-picture kernels works on.  Workgroups take a ticket when they
 * start (one device-scope atomic): ticket t = band t % max_bands of picture t / max_bands.  A band only ever waits for the
 * band above it, which holds a smaller ticket and has therefore STARTED — whatever order the dispatcher chose — so a waiting
 * workgroup can never keep the one it waits for off the machine.  tickets[0] = tickets taken, tickets[1] = workgroups that
 * left: the last one to leave zeroes both for the next launch on this HIP stream (one pair per stream, engine.hip).
 * tickets == nullptr: blockIdx.x is the ticket (single-band launches). */
__device__ __forceinline__ uint32_t take_ticket(uint32_t *tickets, uint32_t *slot)
{
    if (threadIdx.x == 0) *slot = tickets ? atomicAdd(&tickets[0], 1u) : blockIdx.x;
    __syncthreads();
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)*slot);
}
__device__ __forceinline__ void return_ticket(uint32_t *tickets)
{
    if (tickets && threadIdx.x == 0 && atomicAdd(&tickets[1], 1u) == gridDim.x - 1u) { atomicExch(&tickets[0], 0u); atomicExch(&tickets[1], 0u); }
}
/* rows per band and number of bands for a picture of hmb macroblock rows that wants `want` bands, in a launch with
 * max_bands workgroups per picture whose LDS holds the state of at most rows_cap rows */
__device__ __forceinline__ void band_split(int hmb, uint32_t want, uint32_t heavy, uint32_t max_bands, uint32_t light_cap, uint32_t rows_cap, int &rows, int &bands)
{
    if (!heavy && want > light_cap) want = light_cap;         /* (the launch may grant bands to the heavy pictures of a tick only) */
    int w = (int)(want < max_bands ? want : max_bands);
    if (w < 1) w = 1;
    rows = (hmb + w - 1) / w;
    if (rows > (int)rows_cap) rows = (int)rows_cap;
    bands = (hmb + rows - 1) / rows;
}

} // namespace h264k
