/*
 * hd_nal.c — Annex-B byte-stream framing and emulation-prevention removal (H.264 Annex B, 7.4.1).
 *
 * Observable contract kept identical to the reference (src/h264bsd_byte_stream.c:81-237):
 *  - a buffer that starts with 00 00 00 / 00 00 01 is a byte stream, anything else is one raw NAL;
 *  - *read_bytes = leading zeros + start code + payload + trailing zeros that do not belong to the
 *    next start code (at most three zeros are left in front of the next 0x01);
 *  - forbidden byte patterns inside a NAL unit are an error.
 * Unlike the reference the caller's buffer is NOT modified: the unescaped payload is staged in a
 * decoder-owned, zero-padded buffer (which is also what makes the "same pointer, readBytes == 0"
 * re-call protocol cheap).
 */
#include <stdlib.h>
#include <string.h>
#include "hostdec.h"

static int ensure_nal_cap(HostDec *d, uint32_t n)
{
    if (d->nal_cap >= n + 16) return 0;
    uint32_t cap = n + 16 + (n >> 2);
    uint8_t *p = (uint8_t *)realloc(d->nal_buf, cap);
    if (!p) return -1;
    d->nal_buf = p;
    d->nal_cap = cap;
    return 0;
}

int hd_extract_nal(HostDec *d, const uint8_t *s, uint32_t len, uint32_t *read_bytes)
{
    uint32_t start = 0, end = len, consumed = len;
    int raw_nal = 1, invalid = 0;

    if (len > 3 && s[0] == 0 && s[1] == 0 && (s[2] & 0xFE) == 0) {
        raw_nal = 0;
        /* find the first start code prefix: >= 2 zeros followed by 0x01 */
        uint32_t i = 2, zeros = 2;
        for (;;) {
            uint8_t b = s[i++];
            if (i == len) { *read_bytes = len; return -1; }
            if (b == 0) zeros++;
            else if (b == 1 && zeros >= 2) break;
            else zeros = 0;
        }
        start = i;
        /* payload extends to the next start code prefix or the end of the buffer */
        zeros = 0;
        for (;;) {
            uint8_t b = s[i++];
            if (b == 0) zeros++;
            if (b == 1 && zeros >= 2) {
                end = i - zeros - 1;
                uint32_t keep = zeros < 3 ? zeros : 3;   /* zeros owned by the next start code */
                consumed = i - 1 - keep;
                break;
            } else if (b) {
                if (zeros >= 3) invalid = 1;
                zeros = 0;
            }
            if (i == len) { end = len - zeros; consumed = len; break; }
        }
    }
    *read_bytes = consumed;
    if (invalid) return -1;

    uint32_t n = end - start;
    if (ensure_nal_cap(d, n)) return -2;
    uint8_t *w = d->nal_buf;
    const uint8_t *r = s + start;
    uint32_t zeros = 0, out = 0;
    (void)raw_nal;
    for (uint32_t i = 0; i < n; i++) {
        uint8_t b = r[i];
        if (zeros == 2 && b == 3) {
            /* emulation_prevention_three_byte must be followed by 00..03 */
            if (i + 1 == n || r[i + 1] > 3) return -1;
            zeros = 0;
            continue;
        }
        if (zeros == 2 && b <= 2) return -1; /* 000000 / 000001 / 000002 inside a NAL */
        zeros = b ? 0 : zeros + 1;
        w[out++] = b;
    }
    memset(w + out, 0, 16);
    d->nal_size = out;
    return 0;
}
