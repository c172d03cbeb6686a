"""h264bsd_amd — MI355X-native macroblock-reconstruction back end behind the h264bsd C API.

The product is the C-ABI shared library h264bsd_amd/lib/libh264bsd_mi355x.so (host parser in C,
HIP kernels for gfx950).  This package is a thin ctypes mirror of that ABI, named after the reference's
own entry points so tests read like /root/reference/posix/test_h264bsd.c.
"""
from .capi import (  # noqa: F401
    H264BSD_RDY, H264BSD_PIC_RDY, H264BSD_HDRS_RDY, H264BSD_ERROR, H264BSD_PARAM_SET_ERROR, H264BSD_MEMALLOC_ERROR,
    EXPORTED_SYMBOLS, LIB_PATH, FMT_RGBA, FMT_BGRA, FMT_YCBCRA, FMT_I420, Decoder, DevicePicture, BatchDriver, Replay, build, capture_stream, job_header, job_mvs, convert, device_count, device_errors, device_error_events, lib, api_lib, use_product_library, set_tail, pull_batch,
)
