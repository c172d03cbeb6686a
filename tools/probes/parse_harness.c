/* Host parser alone, as a plain C program (capture mode, frame jobs dropped): timing and gprof runs without Python in the
 * way.  Build: see tools/parse_prof.sh.  usage: parse_harness <stream.h264> [passes] */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include <sys/syscall.h>
#include <linux/perf_event.h>
#include <sys/ioctl.h>
static int pe_open(unsigned long long cfg){struct perf_event_attr a; memset(&a,0,sizeof a); a.type=PERF_TYPE_HARDWARE; a.size=sizeof a; a.config=cfg; a.disabled=0; a.exclude_kernel=1; a.exclude_hv=1; return (int)syscall(SYS_perf_event_open,&a,0,-1,-1,0);}
#include "../../include/h264bsd_decoder.h"
#include "../../include/h264bsd_mi355x.h"
int main(int argc, char **argv) {
    FILE *f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    u8 *data = malloc(n), *buf = malloc(n); fread(data, 1, n, f); fclose(f);
    int reps = argc > 2 ? atoi(argv[2]) : 10, pics = 0;
    int fi = pe_open(PERF_COUNT_HW_INSTRUCTIONS), fc = pe_open(PERF_COUNT_HW_CPU_CYCLES); long long i0=0,i1=0,c0=0,c1=0; if(fi>=0) read(fi,&i0,8); if(fc>=0) read(fc,&c0,8);
    struct timespec t0, t1; clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int r = 0; r < reps; r++) {
        memcpy(buf, data, n);
        storage_t *s = h264bsdAlloc();
        h264bsdmiInitCapture(s, 0, NULL, NULL);
        long off = 0;
        while (off < n) { u32 rb = 0; u32 st = h264bsdDecode(s, buf + off, (u32)(n - off), 0, &rb); off += rb; if (st == H264BSD_PIC_RDY) pics++; if (st >= H264BSD_ERROR) break; }
        h264bsdShutdown(s); h264bsdFree(s);
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    double dt = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
    if(fi>=0) read(fi,&i1,8); if(fc>=0) read(fc,&c1,8);
    printf("%d pictures, %.3f ms per picture, %.3f M instructions, %.3f M cycles per picture (perf fd %d %d)\n", pics, dt * 1e3 / pics, (i1-i0)/1e6/pics, (c1-c0)/1e6/pics, fi, fc);
    return 0;
}
