/*
 * engine.h — C interface between the host side (api.c, hostdec) and the HIP engine (engine.hip).
 */
#ifndef H264BSD_AMD_ENGINE_H
#define H264BSD_AMD_ENGINE_H

#include <stdint.h>
#include "hostdec.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Bind a decoder instance to the device engine: fills *sink.  -1 when no HIP device is usable. */
int  eng_attach(JobSink *sink);
/* Stateless colour conversion of a host I420 frame through the GPU (fmt 0 RGBA, 1 BGRA, 2 YCbCrA). */
void eng_convert_host(int fmt, uint32_t width, uint32_t height, const uint8_t *data, uint32_t *out);
/* The HIP device a bound decoder instance lives on (-1: none), and the CPUs of that device's NUMA node (returns how
 * many were written to cpus[]; 0 when the topology cannot be read). */
int  eng_sink_device(const JobSink *sink);
int  eng_device_cpus(int device, int *cpus, int max);

#ifdef __cplusplus
}
#endif
#endif
