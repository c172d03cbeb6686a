"""Decode a byte stream with (a) the compiled reference and (b) this repo's host parser + a pixel back end,
returning comparable traces: the h264bsdDecode call trace and the output pictures in output order.

Back ends for (b): "oracle" (CPU, frame jobs rendered by oracle/pixel_oracle.c — for the -m "not gpu" tests)
or "gpu" (the product: h264bsdInit + HIP engine through the C ABI)."""
import ctypes
import hashlib

import numpy as np

import h264bsd_amd
from h264bsd_amd import capi
from oracle import pyoracle


_libc = ctypes.CDLL(None)
M_PERTURB = -6      # <malloc.h>: every malloc'd block is filled with ~value (glibc)
ZERO_HEAP = 0xFF    # ... so this makes every block the reference mallocs start out as zeros


def reference_is_deterministic(data, no_output_reordering=0):
    """The reference never initialises its frame buffers (src/h264bsd_dpb.c:1026 ALLOCATE, no memset), and on some
    damaged streams it outputs macroblocks it never wrote (e.g. a macroblock whose reconstruction failed but which
    h264bsdMarkSliceCorrupted does not reach), or predicts from them.  Such output is whatever the heap held.
    Detected by decoding twice with glibc's M_PERTURB fill set to two different bytes.  (Informational: the parity
    target on such streams is the reference with zeroed frame buffers, see decode_reference.)"""
    return decode_reference(data, no_output_reordering, 0x55) == decode_reference(data, no_output_reordering, 0xAA)


def decode_reference(data, no_output_reordering=0, heap_fill=ZERO_HEAP):
    """-> (trace, [(sha1 of frame, picId, isIdr, numErrMbs)])

    The compiled reference is run with glibc's M_PERTURB set so that everything it mallocs starts out as zeros: its
    frame buffers are the only memory it reads before writing (every other allocation is memset, src/h264bsd_dpb.c:
    1014-1045, src/h264bsd_storage.c:347-352), a fresh process gets them zeroed from the kernel anyway, and with that
    the reference is a deterministic function of the stream also on the damaged streams that make it show or predict
    from pixels it never wrote.  This repository's frame buffers start out zeroed too (engine.hip sink_configure)."""
    _libc.mallopt(M_PERTURB, heap_fill)
    try:
        return _decode_reference(data, no_output_reordering)
    finally:
        _libc.mallopt(M_PERTURB, 0)


def _decode_reference(data, no_output_reordering):
    ref = pyoracle.RefDecoder()
    lib = ref.lib
    buf = ctypes.create_string_buffer(data, len(data))
    base = ctypes.addressof(buf)
    dec = lib.h264bsdAlloc()
    assert lib.h264bsdInit(dec, no_output_reordering) == 0
    off, trace, pics = 0, [], []
    rb = ctypes.c_uint32(0)
    a, b, c = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32()

    dims = [0, 0]

    def drain():
        if lib.h264bsdPicWidth(dec):       # 0 while a replaced active SPS waits for its activation; pictures of the old one still come out
            dims[:] = [lib.h264bsdPicWidth(dec), lib.h264bsdPicHeight(dec)]
        wmb, hmb = dims
        while True:
            p = lib.h264bsdNextOutputPicture(dec, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
            if not p:
                break
            pics.append((hashlib.sha1(ctypes.string_at(p, wmb * hmb * 384)).hexdigest(), a.value, b.value, c.value))
    pic_id = stall = 0
    while off < len(data):
        r = lib.h264bsdDecode(dec, base + off, len(data) - off, pic_id, ctypes.byref(rb))
        trace.append((int(r), int(rb.value)))
        off += rb.value
        if r == 1:
            pic_id += 1
            drain()
        stall = stall + 1 if rb.value == 0 else 0
        if stall > 3:
            break
    lib.h264bsdFlushBuffer(dec)
    drain()
    lib.h264bsdShutdown(dec)
    lib.h264bsdFree(dec)
    return trace, pics


def decode_ours(data, backend="oracle", no_output_reordering=0):
    pics, trace = [], []
    state = {"dpb": None}

    def on_job(blob):
        if state["dpb"] is None or pyoracle.blob_header(blob)["n_slots"] != len(state["dpb"].slots) or \
                state["dpb"].frame_bytes != pyoracle.blob_header(blob)["n_mbs"] * 384:
            state["dpb"] = pyoracle.OracleDpb(blob)
        state["dpb"].decode(blob)

    dec = capi.Decoder(no_output_reordering, capture=on_job if backend == "oracle" else None)
    buf = ctypes.create_string_buffer(data, len(data))
    base, off = ctypes.addressof(buf), 0

    def drain():
        while True:
            if backend == "oracle":
                o = dec.next_output_info()
                if o is None:
                    break
                slot, pid, idr, nerr = o
                frame = state["dpb"].slots[slot][: state["dpb"].frame_bytes]
            else:
                o = dec.next_output_picture()
                if o is None:
                    break
                frame, pid, idr, nerr = o
            pics.append((hashlib.sha1(np.ascontiguousarray(frame).tobytes()).hexdigest(), pid, idr, nerr))
    pic_id = stall = 0
    while off < len(data):
        r, rb = dec.decode(base + off, len(data) - off, pic_id)
        trace.append((r, rb))
        off += rb
        if r == 1:
            pic_id += 1
            drain()
        elif r == 2:
            # HDRS_RDY: the next slice re-allocates the frame buffers (storage.c:340-378 -> h264bsdResetDpb), also when the
            # new sequence has the same number of macroblocks and buffers as the old one; the product's engine zeroes its
            # buffers at that point (sink_configure), the oracle's stand-in DPB is rebuilt with the next job
            state["dpb"] = None
        stall = stall + 1 if rb == 0 else 0
        if stall > 3:
            break
    dec.flush_buffer()
    drain()
    dec.close()
    return trace, pics
