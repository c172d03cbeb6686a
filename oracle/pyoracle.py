"""ctypes bindings of the TEST-ONLY libraries under oracle/.

  * liboracle.so            - CPU restatement of the pixel path (oracle/pixel_oracle.c)
  * _ref/libh264bsd_ref.so  - the real reference decoder, compiled from /root/reference/src by
                              `make -C oracle ref` (binary only; exists where it was built and on the GPU
                              box, where it travels as a git-ignored artefact)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "liboracle.so")
REF_SO = os.path.join(HERE, "_ref", "libh264bsd_ref.so")


def build(ref=True):
    """Compile the oracle (always) and the reference (only where /root/reference exists)."""
    subprocess.run(["make", "-s", "-C", HERE], check=True)
    if ref and os.path.isdir("/root/reference/src"):
        subprocess.run(["make", "-s", "-C", HERE, "ref"], check=True)


_oracle = None


def oracle_lib():
    global _oracle
    if _oracle is None:
        if not os.path.exists(ORACLE_SO):
            build(ref=False)
        lib = ctypes.CDLL(ORACLE_SO)
        lib.oracle_decode_picture.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]
        lib.oracle_recon.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]
        lib.oracle_deblock.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        lib.oracle_convert.argtypes = [ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]
        lib.oracle_convert.restype = None
        lib.oracle_idct4x4.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        lib.oracle_idct4x4.restype = None
        lib.oracle_luma_dc.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        lib.oracle_luma_dc.restype = None
        lib.oracle_chroma_dc.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        lib.oracle_chroma_dc.restype = None
        lib.oracle_luma_sample.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 6
        lib.oracle_chroma_sample.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 6
        _oracle = lib
    return _oracle


def blob_header(blob):
    """Decode the FjHeader of a frame job: the product's own helper (h264bsd_amd.job_header)."""
    import h264bsd_amd
    return h264bsd_amd.job_header(blob)


class OracleDpb:
    """Host DPB + CPU pixel path: renders frame jobs one after the other."""

    def __init__(self, first_blob):
        h = blob_header(first_blob)
        self.frame_bytes = h["width_mbs"] * h["height_mbs"] * 384
        self.slots = [np.zeros(self.frame_bytes + 64, dtype=np.uint8) for _ in range(h["n_slots"])]
        self._ptrs = (ctypes.c_void_p * 17)(*[s.ctypes.data for s in self.slots])

    def decode(self, blob, deblock=True):
        """Render one picture; returns a view of the slot that received it."""
        lib = oracle_lib()
        buf = ctypes.create_string_buffer(blob, len(blob))
        fn = lib.oracle_decode_picture if deblock else lib.oracle_recon
        if fn(buf, self._ptrs) != 0:
            raise RuntimeError("oracle rejected the frame job")
        return self.slots[blob_header(blob)["cur_slot"]][: self.frame_bytes]


def oracle_convert(fmt, width, height, yuv):
    out = np.empty(width * height, dtype=np.uint32)
    src = np.ascontiguousarray(yuv, dtype=np.uint8)
    oracle_lib().oracle_convert(fmt, width, height, src.ctypes.data, out.ctypes.data)
    return out


def checksum64(frame_u8):
    """Host twin of the device kernel k_checksum (h264bsd_amd/csrc/kernels.hip.h)."""
    w = np.frombuffer(np.ascontiguousarray(frame_u8).tobytes(), dtype="<u4").astype(np.uint64)
    i = np.arange(w.size, dtype=np.uint64)
    mixed = (w ^ ((i * np.uint64(0x9E3779B1)) & np.uint64(0xFFFFFFFF)))
    with np.errstate(over="ignore"):
        return int(np.sum(mixed * (np.uint64(2) * i + np.uint64(1)), dtype=np.uint64))


# ------------------------------------------------------------------ the real reference
class RefDecoder:
    """The compiled reference (h264bsdAlloc/Init/Decode/NextOutputPicture), driven like
    /root/reference/posix/test_h264bsd.c:146-177."""

    def __init__(self):
        if not os.path.exists(REF_SO):
            raise FileNotFoundError(REF_SO)
        lib = ctypes.CDLL(REF_SO)
        lib.h264bsdAlloc.restype = ctypes.c_void_p
        lib.h264bsdInit.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
        lib.h264bsdDecode.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32,
                                      ctypes.POINTER(ctypes.c_uint32)]
        P32 = ctypes.POINTER(ctypes.c_uint32)
        lib.h264bsdNextOutputPicture.restype = ctypes.c_void_p
        lib.h264bsdNextOutputPicture.argtypes = [ctypes.c_void_p, P32, P32, P32]
        for n in ("h264bsdPicWidth", "h264bsdPicHeight", "h264bsdShutdown", "h264bsdFree", "h264bsdFlushBuffer"):
            getattr(lib, n).argtypes = [ctypes.c_void_p]
        for n in ("h264bsdConvertToRGBA", "h264bsdConvertToBGRA", "h264bsdConvertToYCbCrA"):
            getattr(lib, n).argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]
            getattr(lib, n).restype = None
        self.lib = lib

    def decode_stream(self, data, on_frame=None):
        """Decode a whole Annex-B stream.  Returns (trace, n_pics, wmb, hmb); trace = [(ret, readBytes)].
        on_frame(np.uint8 view of the full uncropped I420 frame) is called per output picture."""
        lib = self.lib
        buf = ctypes.create_string_buffer(data, len(data))     # the reference modifies its input
        base = ctypes.addressof(buf)
        dec = lib.h264bsdAlloc()
        assert lib.h264bsdInit(dec, 0) == 0
        off, n, trace = 0, 0, []
        rb = ctypes.c_uint32(0)
        a, b, c = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32()
        wmb = hmb = 0
        while off < len(data):
            r = lib.h264bsdDecode(dec, base + off, len(data) - off, 0, ctypes.byref(rb))
            trace.append((int(r), int(rb.value)))
            off += rb.value
            if r == 1:
                wmb, hmb = lib.h264bsdPicWidth(dec), lib.h264bsdPicHeight(dec)
                while True:
                    p = lib.h264bsdNextOutputPicture(dec, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
                    if not p:
                        break
                    n += 1
                    if on_frame is not None:
                        on_frame(np.frombuffer(ctypes.string_at(p, wmb * hmb * 384), dtype=np.uint8))
            elif r >= 3:
                break
        lib.h264bsdShutdown(dec)
        lib.h264bsdFree(dec)
        return trace, n, wmb, hmb

    def convert(self, fmt, width, height, yuv):
        fn = (self.lib.h264bsdConvertToRGBA, self.lib.h264bsdConvertToBGRA, self.lib.h264bsdConvertToYCbCrA)[fmt]
        src = np.ascontiguousarray(yuv, dtype=np.uint8)
        out = np.empty(width * height, dtype=np.uint32)
        fn(width, height, src.ctypes.data, out.ctypes.data)
        return out
