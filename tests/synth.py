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


def reference_is_deterministic(data, no_output_reordering=0):
    """The reference never initialises its frame buffers (src/h264bsd_dpb.c:1026 ALLOCATE, no memset), and on some
    damaged streams it outputs macroblocks it never wrote (e.g. a macroblock whose reconstruction failed but which
    h264bsdMarkSliceCorrupted does not reach).  Such output is whatever the heap held: undefined behaviour, not a
    parity target.  Detected by decoding twice with glibc's M_PERTURB fill set to two different bytes."""
    outs = []
    for fill in (0x55, 0xAA):
        _libc.mallopt(M_PERTURB, fill)
        try:
            outs.append(decode_reference(data, no_output_reordering))
        finally:
            _libc.mallopt(M_PERTURB, 0)
    return outs[0] == outs[1]


def defined_part_matches(data, ref, ours, no_output_reordering=0):
    """For a stream on which the reference is NOT deterministic: the call trace and every output picture that comes out
    the same from three reference runs (default heap, M_PERTURB 0x55, 0xAA) are still a parity target.
    -> (ok, number of pictures that could be compared)"""
    outs = [ref]
    for fill in (0x55, 0xAA):
        _libc.mallopt(M_PERTURB, fill)
        try:
            outs.append(decode_reference(data, no_output_reordering))
        finally:
            _libc.mallopt(M_PERTURB, 0)
    if not (outs[0][0] == outs[1][0] == outs[2][0]):
        return True, 0                        # even the calls depend on the heap: nothing to compare
    if ours[0] != ref[0] or len(ours[1]) != len(ref[1]):
        return False, 0
    defined = [i for i in range(len(ref[1])) if outs[0][1][i] == outs[1][1][i] == outs[2][1][i]]
    return all(ours[1][i] == ref[1][i] for i in defined), len(defined)


def decode_reference(data, no_output_reordering=0):
    """-> (trace, [(sha1 of frame, picId, isIdr, numErrMbs)])"""
    ref = pyoracle.RefDecoder()
    lib = ref.lib
    buf = ctypes.create_string_buffer(data, len(data))
    base = ctypes.addressof(buf)
    dec = lib.h264bsdAlloc()
    assert lib.h264bsdInit(dec, no_output_reordering) == 0
    off, trace, pics = 0, [], []
    rb = ctypes.c_uint32(0)
    a, b, c = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32()

    def drain():
        wmb, hmb = lib.h264bsdPicWidth(dec), lib.h264bsdPicHeight(dec)
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
        stall = stall + 1 if rb == 0 else 0
        if stall > 3:
            break
    dec.flush_buffer()
    drain()
    dec.close()
    return trace, pics
