/*
 * hd_nal.c — Annex-B byte-stream framing and emulation-prevention removal (H.264 Annex B, 7.4.1).
 *
 * Observable contract kept identical to the reference (src/h264bsd_byte_stream.c:81-237):
 *  - a buffer that starts with 00 00 00 / 00 00 01 is a byte stream, anything else is one raw NAL;
 *  - *read_bytes = leading zeros + start code + payload + trailing zeros that do not belong to the
 *    next start code (at most three zeros are left in front of the next 0x01);
 *  - forbidden byte patterns inside a NAL unit are an error.
 *  - the emulation-prevention bytes are removed IN the caller's buffer as well (the reference compacts the NAL unit
 *    in place, byte_stream.c:193-235, README.md:11: callers that feed a stream twice must keep a private copy, as
 *    the bench harness of SURVEY §8d does), and only when the unit can contain one: a byte-stream unit in which
 *    no 00 00 03 was seen is taken verbatim, without the forbidden-pattern checks (byte_stream.c:127-131, :190).
 * The parser itself reads from a decoder-owned, zero-padded copy of the payload (which is also what makes the "same
 * pointer, readBytes == 0" re-call protocol cheap).
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

int hd_extract_nal(HostDec *d, uint8_t *s, uint32_t len, uint32_t *read_bytes)
{
    uint32_t start = 0, end = len, consumed = len;
    int has_emulation = 1, invalid = 0;                  /* a raw NAL unit is always unescaped (byte_stream.c:181) */

    if (len > 3 && s[0] == 0 && s[1] == 0 && (s[2] & 0xFE) == 0) {
        has_emulation = 0;
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
            if (zeros == 0) {
                /* nothing below looks at a non-zero byte while no zero precedes it: on to the next zero byte at memchr's speed */
                const uint8_t *z = (const uint8_t *)memchr(s + i, 0, len - i);
                if (!z) { end = len; consumed = len; break; }
                i = (uint32_t)(z - s);
            }
            uint8_t b = s[i++];
            if (b == 0) zeros++;
            if (b == 3 && zeros == 2) has_emulation = 1;
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
    uint8_t *r = s + start;
    uint32_t zeros = 0, out = 0;
    int rc = 0;
    if (!has_emulation) {
        memcpy(w, r, n);
        out = n;
    } else {
        for (uint32_t i = 0; i < n; i++) {
            uint8_t b = r[i];
            if (zeros == 2 && b == 3) {
                /* emulation_prevention_three_byte must be followed by 00..03 */
                if (i + 1 == n || r[i + 1] > 3) { rc = -1; break; }
                zeros = 0;
                continue;
            }
            if (zeros == 2 && b <= 2) { rc = -1; break; }   /* 000000 / 000001 / 000002 inside a NAL */
            zeros = b ? 0 : zeros + 1;
            w[out++] = b;
        }
        /* the caller's buffer, compacted as far as the reference gets (src/h264bsd_byte_stream.c unescapes in place and a
         * caller that feeds the same bytes again — after H264BSD_HDRS_RDY the reference consumes nothing — parses the
         * compacted unit): the default, for an identical call trace.  An application that shares one (possibly read-only)
         * stream buffer between decoder instances switches it off (h264bsdmiSetInputReadOnly). */
        if (out < n && !d->input_readonly) memcpy(r, w, out);
        if (rc) return rc;
    }
    memset(w + out, 0, 16);
    d->nal_size = out;
    return 0;
}
