/* kernels/k_pixels_io.hip.h — k_h2d, k_convert*, k_detile, k_output, k_checksum: frame jobs entering, pictures leaving the device.  Part of kernels.hip.h (which see); not a stand-alone header. */
#pragma once
namespace h264k {
/* ------------------------------------------------------------------ frame jobs entering the device */
/* The frame jobs of one tick, fetched from the parser's pinned staging buffers by ONE launch: item i = one job (source in
 * host memory, mapped into the device's address space; destination in the lane's arena).  The host used to enqueue one
 * hipMemcpyAsync per job — 256 runtime calls per tick, 0.7 of the 0.9 ms the enqueueing thread spends between two rounds of
 * parsing (the parser threads wait for it).  H2D_CHUNKS workgroups per job, 16 bytes per lane and trip; sizes are multiples
 * of 32 (FjHeader.total_bytes). */
struct H2dItem { const uint8_t *src; uint8_t *dst; uint32_t bytes, pad; };
constexpr int H2D_CHUNKS = 8;
__global__ __launch_bounds__(256) void k_h2d(const H2dItem *__restrict__ items)
{
    const H2dItem it = items[blockIdx.y];
    const u32x4 *src = reinterpret_cast<const u32x4 *>(it.src);
    u32x4 *dst = reinterpret_cast<u32x4 *>(it.dst);
    const uint32_t n = it.bytes >> 4, stride = gridDim.x * 256u;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += 4u * stride) {
        /* four loads in flight per lane: the link's latency is microseconds */
        const uint32_t i1 = i + stride, i2 = i1 + stride, i3 = i2 + stride;
        const u32x4 a = __builtin_nontemporal_load(src + i);
        u32x4 b = a, c = a, d = a;
        if (i1 < n) b = __builtin_nontemporal_load(src + i1);
        if (i2 < n) c = __builtin_nontemporal_load(src + i2);
        if (i3 < n) d = __builtin_nontemporal_load(src + i3);
        dst[i] = a;
        if (i1 < n) dst[i1] = b;
        if (i2 < n) dst[i2] = c;
        if (i3 < n) dst[i3] = d;
    }
}

/* ------------------------------------------------------------------ pictures leaving the device */
/* The reference's output format is planar I420, uncropped (image.h:46-55).  Frames live in HBM as macroblock tiles,
 * so every path that hands a picture out reads tiles: k_detile (whole frame -> planar), k_output (cropped window ->
 * planar or converted), k_convert with tiled != 0 (whole frame -> RGBA / BGRA / YCbCrA).  k_convert with tiled == 0 is
 * the stateless h264bsdConvertTo*(), whose input is the caller's planar picture. */
__device__ __forceinline__ uint32_t yuv_luma4(const uint8_t *__restrict__ src, int tiled, uint32_t width, uint32_t x, uint32_t y)
{
    return tiled ? *reinterpret_cast<const uint32_t *>(src + luma_at((int)(width >> 4), (int)x, (int)y))
                 : *reinterpret_cast<const uint32_t *>(src + (size_t)y * width + x);
}
__device__ __forceinline__ uint32_t yuv_chroma2(const uint8_t *__restrict__ src, int tiled, uint32_t width, uint32_t height, int plane, uint32_t cx, uint32_t cy)
{
    const uint8_t *p = tiled ? src + chroma_at((int)(width >> 4), plane, (int)cx, (int)cy)
                             : src + (size_t)width * height + (plane ? (size_t)(width >> 1) * (height >> 1) : 0) + (size_t)cy * (width >> 1) + cx;
    return *reinterpret_cast<const uint16_t *>(p);
}
__device__ __forceinline__ uint32_t yuv_pixel(int fmt, int Yv, int cb, int cr)
{
    if (fmt == 2) return 0xFF000000u | ((uint32_t)cr << 16) | ((uint32_t)cb << 8) | (uint32_t)Yv;
    const int c = Yv - 16, d = cb - 128, e = cr - 128;
    const uint32_t r = clip255((298 * c + 409 * e + 128) >> 8);
    const uint32_t g = clip255((298 * c - 100 * d - 208 * e + 128) >> 8);
    const uint32_t b = clip255((298 * c + 516 * d + 128) >> 8);
    return fmt == 0 ? 0xFF000000u | (b << 16) | (g << 8) | r : 0xFF000000u | (r << 16) | (g << 8) | b;
}

/* 4 horizontally adjacent pixels per thread, one 16-byte store; fmt 0 RGBA, 1 BGRA, 2 YCbCrA (bytes in memory order);
 * integer BT.601 limited range, nearest chroma (reference src/h264bsd_decoder.c:1163-1370) */
__global__ __launch_bounds__(256) void k_convert(const uint8_t *__restrict__ yuv, uint32_t *__restrict__ out,
                                                 uint32_t width, uint32_t height, int fmt, size_t in_stride, size_t out_stride, int tiled)
{
    const uint8_t *src = yuv + blockIdx.y * in_stride;
    uint32_t *dst = out + blockIdx.y * out_stride;
    const uint32_t quads_per_row = width >> 2;
    const uint32_t total = quads_per_row * height;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const uint32_t y = i / quads_per_row, x = (i % quads_per_row) * 4;
        const uint32_t yy = yuv_luma4(src, tiled, width, x, y);
        const uint32_t cb2 = yuv_chroma2(src, tiled, width, height, 0, x >> 1, y >> 1), cr2 = yuv_chroma2(src, tiled, width, height, 1, x >> 1, y >> 1);
        uint32_t px[4];
#pragma unroll
        for (int k = 0; k < 4; k++)
            px[k] = yuv_pixel(fmt, (yy >> (8 * k)) & 255, (cb2 >> (8 * (k >> 1))) & 255, (cr2 >> (8 * (k >> 1))) & 255);
        *reinterpret_cast<uint4 *>(dst + (size_t)y * width + x) = make_uint4(px[0], px[1], px[2], px[3]);
    }
}

/* Whole frames, tiles -> 32-bit pixels: one wavefront per PAIR of horizontally adjacent macroblocks (conv_tile_pair,
 * kernels/convert.hip.h: packed 16-bit arithmetic, two rows x four columns per lane).  Same results as k_convert / yuv_pixel
 * (reference decoder.c:1163-1370). */
__global__ __launch_bounds__(256) void k_convert_tiles(const uint8_t *__restrict__ yuv, uint32_t *__restrict__ out, uint32_t wmb, uint32_t hmb,
                                                       int fmt, size_t in_stride, size_t out_stride)
{
    const uint32_t pic = blockIdx.y, chunk = blockIdx.x, n_chunks = gridDim.x;
    const uint8_t *src = yuv + pic * in_stride;
    uint32_t *dst = out + pic * out_stride;
    const uint32_t lane = threadIdx.x & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const ConvPic c = conv_pic(src, dst, wmb);
    const uint32_t n_pairs = c.ppr * hmb;
    const ConvLane cl = conv_lane(c.W, lane);
    uint32_t p = (chunk * 4u + wave) * CONV_BATCH;
    const uint32_t stride = n_chunks * 4u * CONV_BATCH;
    conv_pipeline(c, n_pairs, fmt, cl, lane, [&]() -> uint32_t { const uint32_t r = p; p += stride; return r; });
}

/* The hosted conversion (FrameDesc.conv_src, kernels/convert.hip.h) of a tick that does not launch k_frame_dbk at all (none of its
 * pictures is filtered): the same work as a launch of its own, gridDim.x workgroups per picture. */
__global__ __launch_bounds__(256) void k_convert_rest(const FrameDesc *__restrict__ frames)
{
    const FrameDesc &fd = FD_REF(frames, blockIdx.y);
    if (!fd.conv_src) return;
    const uint32_t lane = threadIdx.x & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const ConvPic c = conv_pic(fd.conv_src, fd.conv_dst, fd.wmb);
    const uint32_t n_pairs = c.ppr * fd.hmb;
    const ConvLane cl = conv_lane(c.W, lane);
    uint32_t p = (blockIdx.x * 4u + wave) * CONV_BATCH;
    const uint32_t stride = gridDim.x * 4u * CONV_BATCH;
    conv_pipeline(c, n_pairs, (int)fd.conv_fmt, cl, lane, [&]() -> uint32_t { const uint32_t r = p; p += stride; return r; });
}

/* Whole frames, tiles -> planar I420 (what h264bsdNextOutputPicture() returns): 16 bytes per thread, a luma row piece
 * or two chroma row pieces of one tile; reads are contiguous per tile, writes 16-byte pieces of planar rows. */
__global__ __launch_bounds__(256) void k_detile(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, uint32_t wmb, uint32_t hmb,
                                                size_t in_stride, size_t out_stride)
{
    const uint8_t *s = src + blockIdx.y * in_stride;
    uint8_t *d = dst + blockIdx.y * out_stride;
    const uint32_t W = wmb * 16, CW = W >> 1;
    const size_t ysz = (size_t)W * hmb * 16, csz = ysz >> 2;
    const uint32_t total = wmb * hmb * (TILE / 16);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const uint32_t mb = i / (TILE / 16), pc = i % (TILE / 16), mbx = mb % wmb, mby = mb / wmb;
        const uint4 v = *reinterpret_cast<const uint4 *>(s + (size_t)i * 16);
        if (pc < 16) *reinterpret_cast<uint4 *>(d + (size_t)(mby * 16 + pc) * W + mbx * 16) = v;
        else {
            const uint32_t plane = (pc - 16) >> 2, r = ((pc - 16) & 3) * 2;     /* two 8-byte chroma rows */
            uint8_t *q = d + ysz + (plane ? csz : 0) + (size_t)(mby * 8 + r) * CW + mbx * 8;
            *reinterpret_cast<uint2 *>(q) = make_uint2(v.x, v.y);
            *reinterpret_cast<uint2 *>(q + CW) = make_uint2(v.z, v.w);
        }
    }
}

/* Device-resident output: the window (x0,y0,w,h) of a decoded frame (even offsets and sizes, multiples of 4 for the
 * window width) either converted (fmt 0..2, tightly packed w*h u32, 4 pixels per lane) or as a tight I420 picture
 * (fmt 3: w*h Y, then the two (w/2)*(h/2) chroma planes; 4 luma samples or 2 chroma samples per lane). */
__global__ __launch_bounds__(256) void k_output(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, uint32_t width,
                                                uint32_t height, int fmt, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h)
{
    const uint32_t qw = w >> 2;
    if (w & 3u) {
        /* window width not a multiple of 4 (cropping is in units of 2 luma samples): one sample / pixel per lane */
        const int twmb = (int)(width >> 4);
        if (fmt == 3) {
            const uint32_t ny = w * h, nc = (w >> 1) * (h >> 1);
            for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < ny + 2 * nc; i += gridDim.x * blockDim.x) {
                if (i < ny) dst[i] = src[luma_at(twmb, (int)(x0 + i % w), (int)(y0 + i / w))];
                else {
                    const uint32_t j = (i - ny) % nc, cw = w >> 1;
                    dst[i] = src[chroma_at(twmb, i - ny >= nc, (int)((x0 >> 1) + j % cw), (int)((y0 >> 1) + j / cw))];
                }
            }
        } else {
            uint32_t *o32 = reinterpret_cast<uint32_t *>(dst);
            for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < w * h; i += gridDim.x * blockDim.x) {
                const uint32_t y = y0 + i / w, x = x0 + i % w;
                o32[i] = yuv_pixel(fmt, src[luma_at(twmb, (int)x, (int)y)], src[chroma_at(twmb, 0, (int)(x >> 1), (int)(y >> 1))],
                                   src[chroma_at(twmb, 1, (int)(x >> 1), (int)(y >> 1))]);
            }
        }
        return;
    }
    if (fmt == 3) {
        const uint32_t ny4 = qw * h, cw = w >> 1, nc2 = (cw >> 1) * (h >> 1);
        for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < ny4 + 2 * nc2; i += gridDim.x * blockDim.x) {
            if (i < ny4) {
                const uint32_t y = i / qw, x = (i % qw) * 4;
                uint32_t v = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) v |= (uint32_t)src[luma_at((int)(width >> 4), (int)(x0 + x + k), (int)(y0 + y))] << (8 * k);
                *reinterpret_cast<uint32_t *>(dst + (size_t)y * w + x) = v;
            } else {
                const uint32_t j = i - ny4, plane = j >= nc2, jj = plane ? j - nc2 : j, y = jj / (cw >> 1), x = (jj % (cw >> 1)) * 2;
                uint32_t v = 0;
#pragma unroll
                for (int k = 0; k < 2; k++) v |= (uint32_t)src[chroma_at((int)(width >> 4), (int)plane, (int)((x0 >> 1) + x + k), (int)((y0 >> 1) + y))] << (8 * k);
                *reinterpret_cast<uint16_t *>(dst + (size_t)w * h + (plane ? (size_t)cw * (h >> 1) : 0) + (size_t)y * cw + x) = (uint16_t)v;
            }
        }
        return;
    }
    (void)height;
    uint32_t *out = reinterpret_cast<uint32_t *>(dst);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < qw * h; i += gridDim.x * blockDim.x) {
        const uint32_t y = y0 + i / qw, x = x0 + (i % qw) * 4;
        uint32_t px[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int Yv = src[luma_at((int)(width >> 4), (int)(x + k), (int)y)];
            const int cb = src[chroma_at((int)(width >> 4), 0, (int)((x + k) >> 1), (int)(y >> 1))];
            const int cr = src[chroma_at((int)(width >> 4), 1, (int)((x + k) >> 1), (int)(y >> 1))];
            px[k] = yuv_pixel(fmt, Yv, cb, cr);
        }
        *reinterpret_cast<uint4 *>(out + (size_t)(i / qw) * w + (i % qw) * 4) = make_uint4(px[0], px[1], px[2], px[3]);
    }
}

/* ------------------------------------------------------------------ on-device verification */
/* sum over the 32-bit words w[i] of the PLANAR picture of (w[i] ^ i*0x9E3779B1) * (2i+1)  (mod 2^64); one block per
 * frame.  The frame is stored as tiles: every 4-byte piece of a tile is one word of the planar picture, whose index i
 * follows from the macroblock position — the value is the one the golden files hold for the reference's output. */
__global__ __launch_bounds__(256) void k_checksum(const uint8_t *__restrict__ base, size_t stride, uint32_t wmb, uint32_t hmb,
                                                  unsigned long long *__restrict__ out)
{
    __shared__ unsigned long long part[256];
    const uint32_t *w = reinterpret_cast<const uint32_t *>(base + blockIdx.x * stride);
    const uint32_t words = wmb * hmb * (TILE / 4), W4 = wmb * 4, CW4 = wmb * 2;
    const uint32_t ywords = W4 * hmb * 16, cwords = ywords >> 2;
    unsigned long long acc = 0;
    for (uint32_t t = threadIdx.x; t < words; t += 256) {
        const uint32_t mb = t / (TILE / 4), k = t % (TILE / 4), mbx = mb % wmb, mby = mb / wmb;
        uint32_t i;
        if (k < 64) i = (mby * 16 + (k >> 2)) * W4 + mbx * 4 + (k & 3);
        else {
            const uint32_t kk = k - 64, plane = kk >> 4, r = (kk & 15) >> 1, half = kk & 1;
            i = ywords + plane * cwords + (mby * 8 + r) * CW4 + mbx * 2 + half;
        }
        acc += (unsigned long long)(w[t] ^ (i * 0x9E3779B1u)) * (unsigned long long)(2u * i + 1u);
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = part[0];
}

} // namespace h264k
