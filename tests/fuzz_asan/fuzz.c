/* ASan/UBSan fuzz harness for the host side (parser, DPB, concealment planning) and the job format: a stream is
 * damaged in seven ways (byte noise, truncation, garbage burst, bit flips near the start, 200 bit flips, a dropped
 * span, one slice NAL unit cut short and repeated up to 400 times — the case that once overran the coefficient
 * section of the frame job), decoded in capture mode, and every frame job is rendered by the (equally instrumented) CPU oracle from an
 * exact-size copy — so an out-of-range slot, coefficient index, list entry or a read past total_bytes is a report.
 * Build + run: tests/fuzz_asan/run.sh <stream.h264> [cases] [seed]     (no GPU needed) */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/h264bsd_mi355x.h"
#include "../../h264bsd_amd/csrc/framejob.h"
int oracle_decode_picture(const uint8_t *blob, uint8_t *const *slots);
static unsigned long long rs=88172645463325252ull;
static unsigned rnd(void){rs^=rs<<13;rs^=rs>>7;rs^=rs<<17;return (unsigned)(rs>>11);}
static uint8_t *slots[17]; static size_t slot_bytes=0; static long rendered=0, rejected=0;
static void cb(void*u,const u8*b,u32 n){(void)u;
  const FjHeader*h=(const FjHeader*)b; size_t fb=(size_t)h->n_mbs*384;
  if(fb!=slot_bytes){for(int i=0;i<17;i++){free(slots[i]);slots[i]=calloc(1,fb);}slot_bytes=fb;}
  u8*copy=malloc(n); memcpy(copy,b,n);           /* exact-size copy: ASan sees reads past total_bytes */
  if(oracle_decode_picture(copy,slots)==0)rendered++;else rejected++;
  free(copy);}
int main(int argc,char**argv){
  FILE*f=fopen(argv[1],"rb");fseek(f,0,SEEK_END);long n=ftell(f);fseek(f,0,SEEK_SET);
  u8*orig=malloc(n);fread(orig,1,n,f);fclose(f);
  int cases=atoi(argv[2]); rs+= (argc>3?atoll(argv[3]):0)*0x9E3779B97F4A7C15ull; long pics=0,errs=0;
  for(int c=0;c<cases;c++){
    long m=n; u8*d=malloc(n); memcpy(d,orig,n);
    int kind=c%7;
    if(kind==0){int k=1+rnd()%40;for(int i=0;i<k;i++)d[rnd()%n]=rnd();}
    else if(kind==1){m=10+rnd()%(n-10);}
    else if(kind==2&&n>700){long s=rnd()%(n-600);for(int i=0;i<500;i++)d[s+i]=rnd();}
    else if(kind==3){for(int i=0;i<8;i++)d[30+rnd()%(n>4100?4000:n-40)]^=1<<(rnd()%8);}
    else if(kind==4){for(int i=0;i<200;i++)d[rnd()%n]^=1<<(rnd()%8);}
    else if(kind==6){ /* one slice NAL unit, cut short, repeated: its macroblocks are decoded, rolled back, decoded again ... */
      long st[4096]; int ns=0; for(long i=0;i+4<n&&ns<4096;i++) if(!d[i]&&!d[i+1]&&!d[i+2]&&d[i+3]==1&&((d[i+4]&31)==1||(d[i+4]&31)==5)) st[ns++]=i;
      if(ns>1){int k=rnd()%(ns-1); long a=st[k], len=st[k+1]-st[k]; for(long j=a+5;j<a+len-3;j++) if(!d[j]&&!d[j+1]&&(d[j+2]==1||(!d[j+2]&&d[j+3]==1))){len=j-a;break;}
        long cut=8+(len>8?rnd()%(len-7):0); if(cut>len)cut=len; int rep=50+rnd()%350;
        u8*x=malloc(a+(size_t)cut*rep+(n-a-len)); memcpy(x,d,a); for(int r=0;r<rep;r++)memcpy(x+a+(size_t)cut*r,d+a,cut);
        memcpy(x+a+(size_t)cut*rep,d+a+len,n-a-len); m=a+(long)cut*rep+(n-a-len); free(d); d=x; }
    }
    else { /* drop a random span (lost packets) */ long s=rnd()%(n-50), l=1+rnd()%(n/10+1); if(s+l>n)l=n-s; memmove(d+s,d+s+l,n-s-l); m=n-l; }
    u8*e=malloc(m); memcpy(e,d,m); free(d);
    storage_t*s=h264bsdAlloc(); h264bsdmiInitCapture(s,0,cb,NULL);
    long off=0;int guard=0,stuck=0;
    while(off<m&&guard++<20000){u32 rb=0;u32 r=h264bsdDecode(s,e+off,(u32)(m-off),0,&rb);
      if(r==1)pics++; if(r>=3)errs++;
      if(rb==0){if(++stuck>3){off++;stuck=0;}}else stuck=0; off+=rb;}
    h264bsdShutdown(s);h264bsdFree(s);free(e);
  }
  printf("cases %d pics %ld errs %ld rendered %ld rejected %ld\n",cases,pics,errs,rendered,rejected);return 0;}
