/* Thread/Address-Sanitizer harness for the THREADED host paths: the parser pool and the three batch calls of include/h264bsd_mi355x.h
 * (decode, pull, pull + decode with the look-ahead of api.c: batch_take), driven like bench.py's end-to-end legs but bound to
 * tests/fuzz_asan/mock_engine.c instead of a GPU.  Every pull must hand out the picture that follows the instance's previous one (the mock's
 * "pixels" are the job's running picture number), in both styles; every instance must consume its whole stream.
 * Build + run: see tests/test_parser_fuzz.py::test_batch_calls_under_thread_sanitizer      usage: batch_tsan <stream.h264> <instances> <threads> */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/h264bsd_mi355x.h"
long mock_jobs(void);
#define MAXN 64
int main(int argc, char **argv)
{
    FILE *f = fopen(argv[1], "rb"); if (!f) return 2;
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    u8 *orig = malloc(n); if (fread(orig, 1, n, f) != (size_t)n) return 2; fclose(f);
    const int N = argc > 2 ? atoi(argv[2]) : 6, T = argc > 3 ? atoi(argv[3]) : 4;
    if (N < 1 || N > MAXN) return 2;
    h264bsdmiSetParserThreads(T);
    long pulls_total = 0;
    for (int style = 0; style < 2; style++) {          /* 0: pull + decode in one call, 1: decode batch, then pull batch */
        storage_t *dec[MAXN]; u8 *buf[MAXN], *cur[MAXN], *pic[MAXN]; u32 len[MAXN], pid[MAXN], status[MAXN], consumed[MAXN], oid[MAXN], oidr[MAXN], onerr[MAXN];
        long pulled[MAXN], decoded[MAXN]; unsigned long long last_seq[MAXN];
        for (int i = 0; i < N; i++) {
            dec[i] = h264bsdAlloc();
            if (h264bsdInit(dec[i], 0) != 0) { printf("init failed\n"); return 1; }
            buf[i] = malloc(n); memcpy(buf[i], orig, n); cur[i] = buf[i]; len[i] = (u32)n; pid[i] = 0; pulled[i] = decoded[i] = 0; last_seq[i] = 0;
        }
        for (int round = 0; round < 100000; round++) {
            int live = 0;
            for (int i = 0; i < N; i++) live += len[i] > 0;
            if (!live) break;
            if (style == 0) {
                if (h264bsdmiPullAndDecodePictureBatch((u32)N, dec, pic, oid, oidr, onerr, cur, len, pid, status, consumed, NULL)) { printf("batch failed\n"); return 1; }
            } else {
                if (h264bsdmiDecodePictureBatch((u32)N, dec, cur, len, pid, status, consumed, NULL)) { printf("batch failed\n"); return 1; }
            }
            for (int i = 0; i < N; i++) {
                cur[i] += consumed[i]; len[i] -= consumed[i];
                if (status[i] == H264BSD_PIC_RDY) { decoded[i]++; pid[i]++; }
            }
            if (style == 1 && h264bsdmiNextOutputPictureBatch((u32)N, dec, pic, oid, oidr, onerr)) { printf("pull failed\n"); return 1; }
            for (int i = 0; i < N; i++) if (pic[i]) {
                unsigned long long seq; memcpy(&seq, pic[i], 8);
                if (pulled[i] && seq != last_seq[i] + 1) { printf("style %d instance %d: picture %llu after %llu\n", style, i, seq, last_seq[i]); return 1; }
                if (oid[i] != (u32)pulled[i]) { printf("style %d instance %d: picId %u, expected %ld\n", style, i, oid[i], pulled[i]); return 1; }
                last_seq[i] = seq; pulled[i]++;
            }
        }
        if (style == 0 && h264bsdmiNextOutputPictureBatch((u32)N, dec, pic, oid, oidr, onerr)) return 1;       /* the last round's pictures */
        for (int i = 0; i < N; i++) {
            if (style == 0 && pic[i]) pulled[i]++;
            if (len[i] || pulled[i] != decoded[i] || !decoded[i]) { printf("style %d instance %d: %u bytes left, %ld decoded, %ld pulled\n", style, i, len[i], decoded[i], pulled[i]); return 1; }
            pulls_total += pulled[i];
            h264bsdShutdown(dec[i]); h264bsdFree(dec[i]); free(buf[i]);
        }
    }
    printf("ok: %d instances x 2 styles, %ld pictures pulled, %ld jobs\n", N, pulls_total, mock_jobs());
    free(orig);
    return 0;
}
