/*
 * A plain C application of the drop-in library, shaped like the reference's own command-line test
 * (posix/test_h264bsd.c:127-183): storage_t on the caller's STACK, one h264bsdDecode() call per NAL unit, pictures
 * pulled with h264bsdNextOutputPicture() after every H264BSD_PIC_RDY, h264bsdShutdown() at the end; `-r N` decodes the
 * file N times over (the reference's -r repeats for ever), `-t N` runs N such decoders on N threads at once.  It is
 * compiled against include/h264bsd_decoder.h only and linked with -lh264bsd_mi355x: nothing of the Python mirror, no
 * ctypes.  For every pass it prints the SHA-256 of the concatenated full (uncropped) I420 frames — the known answers of
 * SURVEY.md §8c.        usage: decode_hash [-r repeats] [-t threads] file.h264
 */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "h264bsd_decoder.h"

/* ---- SHA-256 (FIPS 180-4), enough of it for a test program ---- */
typedef struct { uint32_t h[8]; uint64_t len; uint8_t buf[64]; uint32_t fill; } Sha;
static const uint32_t K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
    0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
    0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
    0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
    0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2 };
#define ROR(x, n) (((x) >> (n)) | ((x) << (32 - (n))))
static void sha_block(Sha *s, const uint8_t *p)
{
    uint32_t w[64], a[8];
    for (int i = 0; i < 16; i++) w[i] = (uint32_t)p[4 * i] << 24 | (uint32_t)p[4 * i + 1] << 16 | (uint32_t)p[4 * i + 2] << 8 | p[4 * i + 3];
    for (int i = 16; i < 64; i++) {
        const uint32_t s0 = ROR(w[i - 15], 7) ^ ROR(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = ROR(w[i - 2], 17) ^ ROR(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    memcpy(a, s->h, sizeof(a));
    for (int i = 0; i < 64; i++) {
        const uint32_t t1 = a[7] + (ROR(a[4], 6) ^ ROR(a[4], 11) ^ ROR(a[4], 25)) + ((a[4] & a[5]) ^ (~a[4] & a[6])) + K[i] + w[i];
        const uint32_t t2 = (ROR(a[0], 2) ^ ROR(a[0], 13) ^ ROR(a[0], 22)) + ((a[0] & a[1]) ^ (a[0] & a[2]) ^ (a[1] & a[2]));
        a[7] = a[6]; a[6] = a[5]; a[5] = a[4]; a[4] = a[3] + t1; a[3] = a[2]; a[2] = a[1]; a[1] = a[0]; a[0] = t1 + t2;
    }
    for (int i = 0; i < 8; i++) s->h[i] += a[i];
}
static void sha_init(Sha *s)
{
    static const uint32_t h0[8] = { 0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19 };
    memcpy(s->h, h0, sizeof(h0)); s->len = 0; s->fill = 0;
}
static void sha_update(Sha *s, const uint8_t *p, size_t n)
{
    s->len += n;
    while (n) {
        if (s->fill == 0 && n >= 64) { sha_block(s, p); p += 64; n -= 64; continue; }
        const size_t k = 64 - s->fill < n ? 64 - s->fill : n;
        memcpy(s->buf + s->fill, p, k); s->fill += (uint32_t)k; p += k; n -= k;
        if (s->fill == 64) { sha_block(s, s->buf); s->fill = 0; }
    }
}
static void sha_hex(Sha *s, char out[65])
{
    const uint64_t bits = s->len * 8;
    uint8_t pad[72] = { 0x80 };
    const size_t padn = (s->fill < 56 ? 56 : 120) - s->fill;
    for (int i = 0; i < 8; i++) pad[padn + i] = (uint8_t)(bits >> (56 - 8 * i));
    sha_update(s, pad, padn + 8);
    for (int i = 0; i < 8; i++) sprintf(out + 8 * i, "%08x", s->h[i]);
}

/* ---- the application ---- */
typedef struct { const u8 *file; size_t size; int repeats, id, pictures, failed; char digest[65]; } Job;

static void decode_content(Job *job)
{
    storage_t dec;                                    /* on the stack, like posix/test_h264bsd.c:129 */
    u8 *content = (u8 *)malloc(job->size);           /* private copy: h264bsdDecode() unescapes the buffer in place */
    for (int pass = 0; pass < job->repeats && !job->failed; pass++) {
        Sha sha;
        sha_init(&sha);
        memcpy(content, job->file, job->size);
        if (h264bsdInit(&dec, HANTRO_FALSE) != HANTRO_OK) { fprintf(stderr, "h264bsdInit failed\n"); job->failed = 1; break; }
        u8 *strm = content;
        u32 len = (u32)job->size, read_bytes = 0, pic_id, is_idr, n_err;
        int pics = 0;
        FILE *dump = NULL;                            /* DH_DUMP=dir: the frames of every decoder, for diagnosing a wrong digest */
        if (getenv("DH_DUMP")) {
            char path[512];
            snprintf(path, sizeof(path), "%s/dec%d_pass%d.yuv", getenv("DH_DUMP"), job->id, pass);
            dump = fopen(path, "wb");
        }
        while (len > 0 && !job->failed) {
            const u32 result = h264bsdDecode(&dec, strm, len, 0, &read_bytes);
            len -= read_bytes;
            strm += read_bytes;
            if (result == H264BSD_PIC_RDY) {
                const u8 *pic;
                while ((pic = h264bsdNextOutputPicture(&dec, &pic_id, &is_idr, &n_err)) != NULL) {
                    sha_update(&sha, pic, (size_t)h264bsdPicWidth(&dec) * h264bsdPicHeight(&dec) * 384u);
                    if (dump) fwrite(pic, 1, (size_t)h264bsdPicWidth(&dec) * h264bsdPicHeight(&dec) * 384u, dump);
                    pics++;
                }
            } else if (result == H264BSD_ERROR || result == H264BSD_PARAM_SET_ERROR || result == H264BSD_MEMALLOC_ERROR) {
                fprintf(stderr, "decoder %d: error %u\n", job->id, result);
                job->failed = 1;
            }
        }
        h264bsdShutdown(&dec);
        if (dump) fclose(dump);
        sha_hex(&sha, job->digest);
        job->pictures = pics;
        printf("decoder %d pass %d: %d pictures sha256 %s\n", job->id, pass, pics, job->digest);
    }
    free(content);
}

static void *thread_main(void *arg) { decode_content((Job *)arg); return NULL; }

int main(int argc, char **argv)
{
    int repeats = 1, threads = 1, i = 1;
    for (; i + 1 < argc && argv[i][0] == '-'; i += 2) {
        if (argv[i][1] == 'r') repeats = atoi(argv[i + 1]);
        else if (argv[i][1] == 't') threads = atoi(argv[i + 1]);
    }
    if (i >= argc || repeats < 1 || threads < 1 || threads > 64) { fprintf(stderr, "usage: decode_hash [-r repeats] [-t threads] file.h264\n"); return 2; }
    FILE *f = fopen(argv[i], "rb");
    if (!f) { perror(argv[i]); return 2; }
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    u8 *file = (u8 *)malloc((size_t)n);
    if (fread(file, 1, (size_t)n, f) != (size_t)n) { fprintf(stderr, "short read\n"); return 2; }
    fclose(f);
    Job jobs[64];
    pthread_t tid[64];
    for (int t = 0; t < threads; t++) {
        jobs[t] = (Job){ file, (size_t)n, repeats, t, 0, 0, "" };
        if (threads == 1) decode_content(&jobs[t]);
        else pthread_create(&tid[t], NULL, thread_main, &jobs[t]);
    }
    int failed = 0;
    for (int t = 0; t < threads; t++) {
        if (threads > 1) pthread_join(tid[t], NULL);
        failed |= jobs[t].failed;
    }
    free(file);
    return failed;
}
