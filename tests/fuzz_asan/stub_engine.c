#include "engine.h"
int eng_attach(JobSink *s){(void)s;return -1;}
void eng_convert_host(int f,uint32_t w,uint32_t h,const uint8_t*d,uint32_t*o){(void)f;(void)w;(void)h;(void)d;(void)o;}
int eng_sink_device(const JobSink *s){(void)s;return -1;}
int eng_device_cpus(int d,int *c,int m){(void)d;(void)c;(void)m;return 0;}
