"""ctypes mirror of include/h264bsd_decoder.h + include/h264bsd_mi355x.h.

No pixel is ever produced in Python or on the CPU here: every call goes through the C ABI of
libh264bsd_mi355x.so, and anything that needs pixels fails loudly when the HIP engine is unavailable.
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.path.join(HERE, "lib", "libh264bsd_mi355x.so")                  # the product: what a C application links
BENCH_LIB_PATH = os.path.join(HERE, "lib", "libh264bsd_mi355x_bench.so")      # same objects + the harness exports; what
                                                                              # this mirror loads (Replay, job tools)
if os.environ.get("H264BSD_VARIANT"):      # A/B experiments: a library built with other kernel flags (tools/experiments/build_variant.sh)
    BENCH_LIB_PATH = os.path.join(HERE, "lib_" + os.environ["H264BSD_VARIANT"], "libh264bsd_mi355x_bench.so")

(H264BSD_RDY, H264BSD_PIC_RDY, H264BSD_HDRS_RDY, H264BSD_ERROR, H264BSD_PARAM_SET_ERROR,
 H264BSD_MEMALLOC_ERROR) = range(6)

# every symbol include/*.h declares (checked by tests/test_abi.py); PRODUCT_SYMBOLS: those of the product library
EXPORTED_SYMBOLS = [
    "h264bsdInit", "h264bsdDecode", "h264bsdShutdown", "h264bsdNextOutputPicture",
    "h264bsdNextOutputPictureRGBA", "h264bsdNextOutputPictureBGRA", "h264bsdNextOutputPictureYCbCrA",
    "h264bsdPicWidth", "h264bsdPicHeight", "h264bsdVideoRange", "h264bsdMatrixCoefficients",
    "h264bsdCroppingParams", "h264bsdSampleAspectRatio", "h264bsdCheckValidParamSets", "h264bsdFlushBuffer",
    "h264bsdProfile", "h264bsdAlloc", "h264bsdFree", "h264bsdConvertToRGBA", "h264bsdConvertToBGRA",
    "h264bsdConvertToYCbCrA",
    "h264bsdmiInitCapture", "h264bsdmiNextOutputInfo", "h264bsdmiNextOutputPictureDevice", "h264bsdmiJobFinalize", "h264bsdmiDeviceCount", "h264bsdmiSetDevice", "h264bsdmiFlush", "h264bsdmiFlushAsync", "h264bsdmiDeviceErrors",
    "h264bsdmiDecodePicture", "h264bsdmiDecodePictureBatch", "h264bsdmiNextOutputPictureBatch", "h264bsdmiPullAndDecodePictureBatch", "h264bsdmiSetParserThreads", "h264bsdmiSetInputReadOnly", "h264bsdmiSetCopyElision",
    "h264bsdmiReplayCreate", "h264bsdmiReplayCreateStaggered", "h264bsdmiReplayCreateDesync", "h264bsdmiReplayCreateSched", "h264bsdmiReplayReschedule", "h264bsdmiReplayDestroy", "h264bsdmiReplayRun", "h264bsdmiReplaySync",
    "h264bsdmiReplayFetch", "h264bsdmiReplayChecksums", "h264bsdmiReplayConvert", "h264bsdmiReplayFetchConverted",
    "h264bsdmiReplayTimings", "h264bsdmiReplaySetConvert", "h264bsdmiReplayConvertTimings", "h264bsdmiReplaySetStages", "h264bsdmiReplaySetTimedKernels", "h264bsdmiReplaySetGroups", "h264bsdmiDebugTailProfile", "h264bsdmiDebugSetTail", "h264bsdmiDebugDeviceErrorEvents", "h264bsdmiReplayJobBytes", "h264bsdmiReplayFrameBytes",
]

class DevicePicture(ctypes.Structure):
    """h264bsdmi_device_picture (include/h264bsd_mi355x.h)"""
    _fields_ = [("data", ctypes.c_void_p), ("width", ctypes.c_uint32), ("height", ctypes.c_uint32), ("pitch", ctypes.c_uint32),
                ("format", ctypes.c_uint32), ("picId", ctypes.c_uint32), ("isIdrPic", ctypes.c_uint32),
                ("numErrMbs", ctypes.c_uint32), ("stream", ctypes.c_void_p)]


FMT_RGBA, FMT_BGRA, FMT_YCBCRA, FMT_I420 = 0, 1, 2, 3

JOB_CB = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint8), ctypes.c_uint32)
P32 = ctypes.POINTER(ctypes.c_uint32)


def build(force=False):
    """Compile the library in-tree for gfx950 (hipcc cross-compiles; no GPU needed)."""
    if force:
        subprocess.run(["make", "-s", "-C", CSRC, "clean"], check=True)
    subprocess.run(["make", "-s", "-C", CSRC], check=True)
    return LIB_PATH


_lib = None


def _share_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64 (same SONAME as /opt/rocm's).  A process that loads this
    library first and imports torch later would end up with torch bound to a runtime it was not built for
    ("No HIP GPUs are available").  When torch is installed, load its runtime first so that both share it — the
    configuration `import torch` before `h264bsd_amd.lib()` gives anyway.  Plain C consumers are unaffected."""
    import importlib.util
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.submodule_search_locations:
        return
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(list(spec.submodule_search_locations)[0], "lib", name)
        if os.path.exists(path):
            try:
                ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
            except OSError:
                return


_api = None
_use_product = False


def use_product_library(on=True):
    """Decoder / BatchDriver / pull_batch / convert / device_* go through libh264bsd_mi355x.so — what a C application links — instead
    of the harness library (same objects plus the replay exports).  tests/test_gpu_api.py switches it on for its module."""
    global _use_product
    _use_product = bool(on)


def api_lib():
    """the library the drop-in API is called through: the product library when use_product_library() is on, else the harness one"""
    global _api
    if not _use_product:
        return lib()
    if _api is None:
        if not os.path.exists(LIB_PATH):
            build()
        _share_torch_hip_runtime()
        _api = _declare(ctypes.CDLL(LIB_PATH), harness=False)
    return _api


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH) or not os.path.exists(BENCH_LIB_PATH):
        build()
    _share_torch_hip_runtime()
    _lib = _declare(ctypes.CDLL(BENCH_LIB_PATH), harness=True)
    return _lib


def _declare(L, harness):
    vp, u32, u8p = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p
    L.h264bsdAlloc.restype = vp
    L.h264bsdFree.argtypes = [vp]
    L.h264bsdInit.argtypes = [vp, u32]
    L.h264bsdInit.restype = u32
    L.h264bsdmiInitCapture.argtypes = [vp, u32, JOB_CB, vp]
    L.h264bsdmiInitCapture.restype = u32
    L.h264bsdDecode.argtypes = [vp, u8p, u32, u32, P32]
    L.h264bsdDecode.restype = u32
    L.h264bsdShutdown.argtypes = [vp]
    for n in ("h264bsdNextOutputPicture", "h264bsdNextOutputPictureRGBA", "h264bsdNextOutputPictureBGRA",
              "h264bsdNextOutputPictureYCbCrA"):
        getattr(L, n).argtypes = [vp, P32, P32, P32]
        getattr(L, n).restype = vp
    for n in ("h264bsdPicWidth", "h264bsdPicHeight", "h264bsdVideoRange", "h264bsdMatrixCoefficients",
              "h264bsdCheckValidParamSets", "h264bsdProfile"):
        getattr(L, n).argtypes = [vp]
        getattr(L, n).restype = u32
    L.h264bsdFlushBuffer.argtypes = [vp]
    L.h264bsdCroppingParams.argtypes = [vp, P32, P32, P32, P32, P32]
    L.h264bsdSampleAspectRatio.argtypes = [vp, P32, P32]
    for n in ("h264bsdConvertToRGBA", "h264bsdConvertToBGRA", "h264bsdConvertToYCbCrA"):
        getattr(L, n).argtypes = [u32, u32, vp, vp]
        getattr(L, n).restype = None
    L.h264bsdmiNextOutputInfo.argtypes = [vp, P32, P32, P32]
    L.h264bsdmiNextOutputInfo.restype = ctypes.c_int
    L.h264bsdmiNextOutputPictureDevice.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(DevicePicture)]
    L.h264bsdmiNextOutputPictureDevice.restype = ctypes.c_int
    L.h264bsdmiDecodePicture.argtypes = [vp, u8p, u32, u32, P32, P32]
    L.h264bsdmiDecodePicture.restype = u32
    L.h264bsdmiDecodePictureBatch.argtypes = [u32, ctypes.POINTER(vp), ctypes.POINTER(vp), P32, P32, P32, P32, P32]
    L.h264bsdmiPullAndDecodePictureBatch.argtypes = [u32, ctypes.POINTER(vp), ctypes.POINTER(vp), P32, P32, P32, ctypes.POINTER(vp), P32, P32, P32, P32, P32]
    L.h264bsdmiSetParserThreads.argtypes = [ctypes.c_int]
    L.h264bsdmiSetInputReadOnly.argtypes = [vp, u32]
    L.h264bsdmiSetCopyElision.argtypes = [vp, u32]
    L.h264bsdmiSetCopyElision.restype = ctypes.c_int
    L.h264bsdmiSetDevice.argtypes = [ctypes.c_int]
    L.h264bsdmiDeviceErrors.restype = ctypes.c_uint
    L.h264bsdmiNextOutputPictureBatch.argtypes = [u32, ctypes.POINTER(vp), ctypes.POINTER(vp), P32, P32, P32]
    if not harness:
        return L
    L.h264bsdmiJobFinalize.argtypes = [ctypes.c_void_p, u32, u32]
    L.h264bsdmiReplayCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p), P32, u32, u32]
    L.h264bsdmiReplayCreate.restype = vp
    L.h264bsdmiReplayCreateStaggered.argtypes = [ctypes.POINTER(ctypes.c_void_p), P32, u32, u32, u32]
    L.h264bsdmiReplayCreateStaggered.restype = vp
    L.h264bsdmiReplayCreateDesync.argtypes = [ctypes.POINTER(ctypes.c_void_p), P32, u32, u32, P32, u32, u32]
    L.h264bsdmiReplayCreateDesync.restype = vp
    L.h264bsdmiReplayCreateSched.argtypes = [ctypes.POINTER(ctypes.c_void_p), P32, u32, u32, P32, u32, u32, u32]
    L.h264bsdmiReplayCreateSched.restype = vp
    L.h264bsdmiReplayReschedule.argtypes = [vp, P32, u32, u32, u32]
    L.h264bsdmiReplayDestroy.argtypes = [vp]
    L.h264bsdmiReplayDestroy.restype = None
    L.h264bsdmiReplayRun.argtypes = [vp, u32, u32]
    L.h264bsdmiReplaySync.argtypes = [vp]
    L.h264bsdmiReplayFetch.argtypes = [vp, u32, u32, vp]
    L.h264bsdmiReplayChecksums.argtypes = [vp, u32, vp]
    L.h264bsdmiReplayConvert.argtypes = [vp, u32, ctypes.c_int]
    L.h264bsdmiReplayFetchConverted.argtypes = [vp, u32, vp]
    L.h264bsdmiReplayTimings.argtypes = [vp, ctypes.POINTER(ctypes.c_float), P32]
    L.h264bsdmiReplaySetConvert.argtypes = [vp, ctypes.c_int]
    L.h264bsdmiReplayConvertTimings.argtypes = [vp, ctypes.POINTER(ctypes.c_float), P32]
    L.h264bsdmiReplaySetStages.argtypes = [vp, ctypes.c_uint]
    L.h264bsdmiReplaySetTimedKernels.argtypes = [vp, ctypes.c_uint]
    L.h264bsdmiReplaySetGroups.argtypes = [vp, u32]
    L.h264bsdmiDebugSetTail.argtypes = [u32] * 7
    L.h264bsdmiReplayJobBytes.argtypes = [vp]
    L.h264bsdmiReplayJobBytes.restype = ctypes.c_ulonglong
    L.h264bsdmiReplayFrameBytes.argtypes = [vp]
    L.h264bsdmiReplayFrameBytes.restype = u32
    return L


def device_count():
    return int(api_lib().h264bsdmiDeviceCount())


KEEP = 0xFFFFFFFF


def set_tail(dbk_rows_light=KEEP, dbk_rows_heavy=KEEP, dbk_waves=KEEP, intra_rows_light=KEEP, intra_rows_heavy=KEEP, intra_waves=KEEP,
             band_budget=KEEP):
    """How the per-picture kernels split the pictures of replay sets / decoders created from now on into row bands
    (h264bsdmiDebugSetTail): rows per band for light / heavy pictures (0 = one band) and wavefronts per workgroup."""
    lib().h264bsdmiDebugSetTail(dbk_rows_light, dbk_rows_heavy, dbk_waves, intra_rows_light, intra_rows_heavy, intra_waves, band_budget)


def device_errors():
    """sticky DEVERR_* bits of the engine (0 = none): tripwires of the kernels that must never fire"""
    return int(api_lib().h264bsdmiDeviceErrors())


def device_error_events():
    """how often a tripwire of the kernels has fired since the library was loaded (monotonic; bench library)"""
    L = lib()
    L.h264bsdmiDebugDeviceErrorEvents.restype = ctypes.c_uint
    return int(L.h264bsdmiDebugDeviceErrorEvents())


class Decoder:
    """One decoder instance.  Method names follow the reference API (h264bsd_decoder.h)."""

    def __init__(self, no_output_reordering=0, capture=None, copy_elision=None):
        """capture: None -> pixels on the GPU (h264bsdInit; raises when there is no device);
        a callable(bytes) -> parser only, every picture's frame job is handed to it.
        copy_elision: None -> the library's default (on with a device, off in capture mode), else h264bsdmiSetCopyElision."""
        L = api_lib()
        self._L = L
        self._st = L.h264bsdAlloc()
        self._cb = None
        if capture is None:
            rc = L.h264bsdInit(self._st, no_output_reordering)
        elif capture == "discard":           # parser only, frame jobs dropped (parser benchmarks)
            rc = L.h264bsdmiInitCapture(self._st, no_output_reordering, ctypes.cast(None, JOB_CB), None)
        else:
            self._cb = JOB_CB(lambda user, p, n: capture(ctypes.string_at(p, n)))
            rc = L.h264bsdmiInitCapture(self._st, no_output_reordering, self._cb, None)
        if rc != 0:
            L.h264bsdFree(self._st)
            self._st = None
            raise RuntimeError("h264bsdInit failed: the HIP engine is not available (no CPU pixel path exists)")
        if copy_elision is not None:
            L.h264bsdmiSetCopyElision(self._st, 1 if copy_elision else 0)

    def close(self):
        if self._st:
            self._L.h264bsdShutdown(self._st)
            self._L.h264bsdFree(self._st)
            self._st = None

    __del__ = close

    def decode(self, buf_addr, length, pic_id=0):
        """h264bsdDecode on raw memory: returns (status, readBytes)."""
        rb = ctypes.c_uint32(0)
        r = self._L.h264bsdDecode(self._st, buf_addr, length, pic_id, ctypes.byref(rb))
        return int(r), int(rb.value)

    def _next(self, fn, nbytes, dtype):
        a, b, c = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32()
        p = fn(self._st, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
        if not p:
            return None
        arr = np.frombuffer(ctypes.string_at(p, nbytes), dtype=dtype)
        return arr, int(a.value), int(b.value), int(c.value)

    def next_output_info(self):
        """(slot, picId, isIdr, numErrMbs) of the next output picture, or None (works in capture mode)"""
        a, b, c = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32()
        slot = self._L.h264bsdmiNextOutputInfo(self._st, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
        return None if slot < 0 else (slot, int(a.value), int(b.value), int(c.value))

    def next_output_picture_device(self, fmt=FMT_I420, crop=False):
        """Pop the next output picture and leave it in HBM.  Returns (tensor, picId, isIdr, numErrMbs) or None;
        tensor is a zero-copy torch.uint8 CUDA tensor: [h*3/2, w] for I420 (Y rows, then Cb and Cr rows of w/2 bytes
        packed back to back), [h, w, 4] for the converted formats.  It aliases decoder memory: valid until the next
        decode()/close() of this instance — clone() to keep it."""
        import torch
        pic = DevicePicture()
        rc = self._L.h264bsdmiNextOutputPictureDevice(self._st, fmt, 1 if crop else 0, ctypes.byref(pic))
        if rc == 0:
            return None
        if rc < 0:
            raise RuntimeError(f"h264bsdmiNextOutputPictureDevice failed ({rc})")
        n = pic.width * pic.height * 3 // 2 if fmt == FMT_I420 else pic.width * pic.height * 4

        class _Buf:
            __cuda_array_interface__ = {"shape": (n,), "typestr": "|u1", "data": (pic.data, False), "version": 2}
        t = torch.as_tensor(_Buf(), device="cuda")
        t = t.view(pic.height * 3 // 2, pic.width) if fmt == FMT_I420 else t.view(pic.height, pic.width, 4)
        return t, int(pic.picId), int(pic.isIdrPic), int(pic.numErrMbs)

    def frame_bytes(self):
        return self.pic_width() * self.pic_height() * 384

    def next_output_picture(self):
        """(uint8 I420 frame copy, picId, isIdr, numErrMbs) or None"""
        return self._next(self._L.h264bsdNextOutputPicture, self.frame_bytes(), np.uint8)

    def next_output_picture_converted(self, fmt):
        fn = (self._L.h264bsdNextOutputPictureRGBA, self._L.h264bsdNextOutputPictureBGRA,
              self._L.h264bsdNextOutputPictureYCbCrA)[fmt]
        return self._next(fn, self.pic_width() * self.pic_height() * 256 * 4, np.uint32)

    def pic_width(self):
        return int(self._L.h264bsdPicWidth(self._st))

    def pic_height(self):
        return int(self._L.h264bsdPicHeight(self._st))

    def video_range(self):
        return int(self._L.h264bsdVideoRange(self._st))

    def matrix_coefficients(self):
        return int(self._L.h264bsdMatrixCoefficients(self._st))

    def profile(self):
        return int(self._L.h264bsdProfile(self._st))

    def check_valid_param_sets(self):
        return int(self._L.h264bsdCheckValidParamSets(self._st))

    def flush_buffer(self):
        self._L.h264bsdFlushBuffer(self._st)

    def cropping_params(self):
        v = [ctypes.c_uint32() for _ in range(5)]
        self._L.h264bsdCroppingParams(self._st, *[ctypes.byref(x) for x in v])
        return tuple(int(x.value) for x in v)   # flag, left, width, top, height

    def sample_aspect_ratio(self):
        w, h = ctypes.c_uint32(), ctypes.c_uint32()
        self._L.h264bsdSampleAspectRatio(self._st, ctypes.byref(w), ctypes.byref(h))
        return int(w.value), int(h.value)

    def decode_stream(self, data, on_picture=None, drain=True):
        """The reference harness loop (posix/test_h264bsd.c:146-177) over a whole byte stream.
        Returns the call trace [(status, readBytes)]."""
        buf = ctypes.create_string_buffer(data, len(data))
        base, off, trace = ctypes.addressof(buf), 0, []
        while off < len(data):
            r, rb = self.decode(base + off, len(data) - off)
            trace.append((r, rb))
            off += rb
            if r == H264BSD_PIC_RDY and drain:
                while True:
                    pic = self.next_output_picture() if on_picture is not None else None
                    if pic is None:
                        break
                    on_picture(*pic)
            elif r >= H264BSD_ERROR:
                break
        return trace


class BatchDriver:
    """Advance many Decoder instances picture by picture on the library's parser threads
    (h264bsdmiDecodePictureBatch): one call parses the next picture of every stream that still has data."""

    def __init__(self, decoders, streams):
        """streams: one bytes object per decoder (each is copied once into a private buffer)"""
        self.L = api_lib()
        self.n = len(decoders)
        self.decoders = decoders
        self._bufs = [ctypes.create_string_buffer(b, len(b)) for b in streams]
        self.size = [len(b) for b in streams]
        self.off = [0] * self.n
        self.pictures = [0] * self.n
        VP = ctypes.c_void_p * self.n
        U32 = ctypes.c_uint32 * self.n
        self._dec = VP(*[d._st for d in decoders])
        self._ptr, self._len, self._pid = VP(), U32(), U32()
        self._status, self._consumed, self._errs = U32(), U32(), U32()
        self._out, self._oid, self._oidr, self._onerr = VP(), U32(), U32(), U32()
        self.pulled = {}
        self._draining = set()

    def step(self, pull=False):
        """parse the next picture of every unfinished stream; returns the indices that produced a picture.
        pull=True: h264bsdmiPullAndDecodePictureBatch — every stream's next output picture is pulled first (self.pulled: stream index ->
        (host pointer, picId, isIdrPic, numErrMbs) of the streams that had one), on the same threads, beside the parsing of the other streams."""
        live = [k for k in range(self.n) if self.off[k] < self.size[k]]
        self.pulled = {}
        if pull:
            # a stream that has consumed its input stays in the batch with len = 0 ("pull only", include/h264bsd_mi355x.h) until a
            # pull comes back empty: the pictures still in its output queue are not silently dropped
            live = live + [k for k in sorted(self._draining) if k not in live]
        if not live:
            return []
        if pull:
            for i, k in enumerate(live):
                self._dec[i] = self.decoders[k]._st
                self._ptr[i] = ctypes.addressof(self._bufs[k]) + min(self.off[k], self.size[k] - 1)
                self._len[i] = self.size[k] - self.off[k]
                self._pid[i] = self.pictures[k]
            rc = self.L.h264bsdmiPullAndDecodePictureBatch(len(live), self._dec, self._out, self._oid, self._oidr, self._onerr,
                                                           self._ptr, self._len, self._pid, self._status, self._consumed, self._errs)
            if rc != 0:
                raise RuntimeError("h264bsdmiPullAndDecodePictureBatch failed")
            ready = []
            for i, k in enumerate(live):
                if self._out[i]:
                    self.pulled[k] = (self._out[i], int(self._oid[i]), int(self._oidr[i]), int(self._onerr[i]))
                elif k in self._draining:
                    self._draining.discard(k)                  # its queue is empty: done
                self.off[k] += self._consumed[i]
                if self.off[k] >= self.size[k] and k not in self._draining and (self._out[i] or self._status[i] == H264BSD_PIC_RDY):
                    self._draining.add(k)                      # input finished: come back for what is still queued
                if self._status[i] == H264BSD_PIC_RDY:
                    self.pictures[k] += 1
                    ready.append(k)
            return ready
        for i, k in enumerate(live):
            self._dec[i] = self.decoders[k]._st
            self._ptr[i] = ctypes.addressof(self._bufs[k]) + self.off[k]
            self._len[i] = self.size[k] - self.off[k]
            self._pid[i] = self.pictures[k]
        rc = self.L.h264bsdmiDecodePictureBatch(len(live), self._dec, self._ptr, self._len, self._pid, self._status,
                                                self._consumed, self._errs)
        if rc != 0:
            raise RuntimeError("h264bsdmiDecodePictureBatch failed")
        ready = []
        for i, k in enumerate(live):
            self.off[k] += self._consumed[i]
            if self._status[i] == H264BSD_PIC_RDY:
                self.pictures[k] += 1
                ready.append(k)
        return ready


def pull_batch(decoders, frame_bytes=None):
    """h264bsdmiNextOutputPictureBatch: the next output picture of every decoder, pulled on the library's threads.
    Returns (pointers, pic ids); with frame_bytes also numpy views of the pinned host pictures (valid until the next decode call)."""
    import numpy as np
    L = api_lib()
    n = len(decoders)
    VP = ctypes.c_void_p * n
    U32 = ctypes.c_uint32 * n
    dec = VP(*[d._st for d in decoders])
    out, ids, idr, nerr = VP(), U32(), U32(), U32()
    if L.h264bsdmiNextOutputPictureBatch(n, dec, out, ids, idr, nerr) != 0:
        raise RuntimeError("h264bsdmiNextOutputPictureBatch failed")
    ptrs = [out[i] for i in range(n)]
    if frame_bytes is None:
        return ptrs, list(ids)
    views = [None if not p else np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint8)), (frame_bytes,)) for p in ptrs]
    return views, list(ids)


def job_header(blob):
    """Decode the FjHeader of a frame job (h264bsd_amd/csrc/framejob.h)."""
    import struct
    (magic, total, wmb, hmb, n_mbs, cur, is_idr, n_slots, any_dbk, rec_off, mv_off, lvl_off, idx_off, coef_off,
     n_intra, n_levels, n_coef, n_inter, pic_seq, copy_off, n_copy, gen_off, n_gen, dbk_off, n_dbk, n_gen_uniform, ghost,
     dbk_only, _down, n_gen_quad, mvx_off, n_mvx) = struct.unpack_from("<IIHHIBBBBIIIIIIIIIIIIIIIIIIIIIII", blob, 0)
    if magic != 0x314A4648:
        raise ValueError("not a frame job")
    # FjCopy entries are 8 bytes {u16 mb, u8 slot, u8 count, i16 dx, i16 dy}: macroblocks moved by k_copy
    n_copy_mbs = sum(blob[copy_off + 8 * i + 3] for i in range(n_copy))
    return dict(total_bytes=total, width_mbs=wmb, height_mbs=hmb, n_mbs=n_mbs, cur_slot=cur, is_idr=is_idr,
                n_slots=n_slots, any_deblock=any_dbk, rec_off=rec_off, mv_off=mv_off, lvl_off=lvl_off,
                idx_off=idx_off, coef_off=coef_off, n_intra=n_intra, n_intra_levels=n_levels,
                n_coef_blocks=n_coef, n_inter=n_inter, pic_seq=pic_seq, copy_off=copy_off, n_copy=n_copy,
                gen_off=gen_off, n_gen=n_gen, dbk_off=dbk_off, n_dbk=n_dbk, n_copy_mbs=n_copy_mbs,
                n_gen_uniform=n_gen_uniform, n_gen_quad=n_gen_quad, ghost=ghost, dbk_only=dbk_only, mvx_off=mvx_off, n_mvx=n_mvx)


def job_mvs(blob):
    """The motion vectors of a FINISHED frame job as a dense int16 array [n_mbs][16][2] (raster 4x4 order, quarter samples):
    a macroblock with FJ_PRED_UNIFORM_MV (pred bit 6) carries its one vector in the record (bytes 24..27), the other inter
    macroblocks have sixteen in the sparse section at mvx_off, entry = u32 at bytes 28..31 of the record (framejob.h)."""
    h = job_header(blob)
    n = h["n_mbs"]
    rec = np.frombuffer(blob, dtype=np.uint8, count=n * 32, offset=h["rec_off"]).reshape(n, 32)
    mvx = np.frombuffer(blob, dtype=np.int16, count=h["n_mvx"] * 32, offset=h["mvx_off"]).reshape(h["n_mvx"], 16, 2)
    out = np.zeros((n, 16, 2), dtype=np.int16)
    inter = rec[:, 0] == 0
    one = inter & ((rec[:, 4] & 0x40) != 0)
    mv1 = np.frombuffer(rec[:, 24:28].tobytes(), dtype=np.int16).reshape(n, 2)
    idx = np.frombuffer(rec[:, 28:32].tobytes(), dtype=np.uint32)
    out[one] = mv1[one][:, None, :]
    many = inter & ~one
    out[many] = mvx[idx[many]]
    return out


def capture_stream(data, copy_elision=False):
    """Parse a byte stream on the host only.  Returns (jobs, trace, info): the packed frame job of every
    picture in decode order, the h264bsdDecode call trace, and stream geometry.  copy_elision: leave out the copies that
    would rewrite what the destination frame buffer holds, like a decoder bound to a device does — such jobs are only
    good for an in-order replay from an IDR picture onto persistent frames (Replay), not for rendering a picture alone."""
    jobs = []
    dec = Decoder(capture=jobs.append, copy_elision=copy_elision)
    trace = dec.decode_stream(data, drain=False)
    info = dict(width_mbs=dec.pic_width(), height_mbs=dec.pic_height(), cropping=dec.cropping_params(),
                video_range=dec.video_range(), matrix_coefficients=dec.matrix_coefficients(),
                profile=dec.profile(), sar=dec.sample_aspect_ratio())
    dec.close()
    return jobs, trace, info


def convert(fmt, width, height, yuv):
    """h264bsdConvertToRGBA/BGRA/YCbCrA (fmt 0/1/2) of one host I420 frame, computed on the GPU."""
    if device_count() <= 0:
        raise RuntimeError("h264bsdConvertTo*: no HIP device (no CPU pixel path exists)")
    L = api_lib()
    fn = (L.h264bsdConvertToRGBA, L.h264bsdConvertToBGRA, L.h264bsdConvertToYCbCrA)[fmt]
    src = np.ascontiguousarray(yuv, dtype=np.uint8)
    out = np.zeros(width * height, dtype=np.uint32)
    fn(width, height, src.ctypes.data, out.ctypes.data)
    return out


class Replay:
    """HBM-resident replay set: n_streams private copies of one captured stream (kernels only)."""

    def __init__(self, jobs, n_streams, odd_offset=0, offsets=None, heavy_lanes=0, heavy_delay=4, groups=1):
        """odd_offset: odd-numbered streams run picture (k + odd_offset) % n_pics in tick k ("staggered").
        offsets: first picture of every stream (desynchronised streams); heavy_lanes > 0: mostly intra-coded pictures
        run on extra HIP streams and rejoin heavy_delay ticks later (run() then always runs one whole lap); groups > 1
        (with heavy_lanes > 0): the streams are split into groups that run their own ticks on their own HIP streams."""
        L = lib()
        self._L = L
        self._keep = [ctypes.create_string_buffer(j, len(j)) for j in jobs]
        ptrs = (ctypes.c_void_p * len(jobs))(*[ctypes.addressof(b) for b in self._keep])
        sizes = (ctypes.c_uint32 * len(jobs))(*[len(j) for j in jobs])
        self.n_pics, self.n_streams = len(jobs), n_streams
        self.odd_offset = odd_offset
        if offsets is None:
            offsets = [odd_offset if s & 1 else 0 for s in range(n_streams)]
        self.offsets = [int(o) for o in offsets]
        offs = (ctypes.c_uint32 * n_streams)(*self.offsets)
        self._h = L.h264bsdmiReplayCreateSched(ptrs, sizes, len(jobs), n_streams, offs, heavy_lanes, heavy_delay, groups)
        if not self._h:
            raise RuntimeError("h264bsdmiReplayCreate failed (no HIP device or out of memory)")
        self.frame_bytes = int(L.h264bsdmiReplayFrameBytes(self._h))
        self.job_bytes = int(L.h264bsdmiReplayJobBytes(self._h))

    def reschedule(self, odd_offset=0, offsets=None, heavy_lanes=0, heavy_delay=4, groups=1):
        """Another schedule for the same resident jobs (arguments as in __init__): no second allocation and upload.
        Frame buffers start from zero again."""
        if offsets is None:
            offsets = [odd_offset if s & 1 else 0 for s in range(self.n_streams)]
        self.odd_offset = odd_offset
        self.offsets = [int(o) for o in offsets]
        offs = (ctypes.c_uint32 * self.n_streams)(*self.offsets)
        if self._L.h264bsdmiReplayReschedule(self._h, offs, heavy_lanes, heavy_delay, groups) != 0:
            raise RuntimeError("h264bsdmiReplayReschedule failed")

    def close(self):
        if self._h:
            self._L.h264bsdmiReplayDestroy(self._h)
            self._h = None

    __del__ = close

    def run(self, first=0, count=None):
        count = self.n_pics - first if count is None else count
        if self._L.h264bsdmiReplayRun(self._h, first, count) != 0:
            raise RuntimeError("h264bsdmiReplayRun failed")

    def sync(self):
        if self._L.h264bsdmiReplaySync(self._h) != 0:
            raise RuntimeError("h264bsdmiReplaySync failed")

    def fetch(self, stream, slot):
        out = np.empty(self.frame_bytes, dtype=np.uint8)
        if self._L.h264bsdmiReplayFetch(self._h, stream, slot, out.ctypes.data) != 0:
            raise RuntimeError("h264bsdmiReplayFetch failed")
        return out

    def checksums(self, slot):
        out = np.empty(self.n_streams, dtype=np.uint64)
        if self._L.h264bsdmiReplayChecksums(self._h, slot, out.ctypes.data) != 0:
            raise RuntimeError("h264bsdmiReplayChecksums failed")
        return out

    def set_convert(self, fmt, trailing=True, hosting=True, conv_waves=0):
        """fmt 0..2: every picture run() produces is converted (hosted by the next tick's k_frame_dbk where possible, a launch of
        its own otherwise); -1: off.  trailing=False: no launch behind the last tick of a run; hosting=False: launches only;
        conv_waves: conversion wavefronts per k_frame_dbk workgroup (0 = the default)."""
        if fmt >= 0:
            fmt |= (0 if trailing else 0x100) | (0 if hosting else 0x200) | ((conv_waves & 15) << 16)
        if self._L.h264bsdmiReplaySetConvert(self._h, fmt) != 0:
            raise RuntimeError("h264bsdmiReplaySetConvert failed")

    def convert_timings(self):
        ms, n = ctypes.c_float(0), ctypes.c_uint32(0)
        if self._L.h264bsdmiReplayConvertTimings(self._h, ctypes.byref(ms), ctypes.byref(n)) != 0:
            raise RuntimeError("h264bsdmiReplayConvertTimings failed")
        return ms.value, n.value

    def convert(self, slot, fmt):
        if self._L.h264bsdmiReplayConvert(self._h, slot, fmt) != 0:
            raise RuntimeError("h264bsdmiReplayConvert failed")

    def fetch_converted(self, stream, n_pixels):
        out = np.empty(n_pixels, dtype=np.uint32)
        if self._L.h264bsdmiReplayFetchConverted(self._h, stream, out.ctypes.data) != 0:
            raise RuntimeError("h264bsdmiReplayFetchConverted failed")
        return out

    def set_groups(self, n):
        if self._L.h264bsdmiReplaySetGroups(self._h, n) != 0:
            raise RuntimeError('h264bsdmiReplaySetGroups failed')

    def set_stages(self, mask):
        self._L.h264bsdmiReplaySetStages(self._h, mask)

    def set_timed_kernels(self, mask):
        """bit k: KERNELS[k] is bracketed by HIP events in the following run() calls (default all)"""
        self._timed_mask = mask & 31
        self._L.h264bsdmiReplaySetTimedKernels(self._h, mask)

    KERNELS = ("k_copy", "k_recon_inter", "k_dbk", "k_frame_intra", "k_frame_dbk")

    def timings(self):
        """HIP-event times of the last run(): {kernel: (ms, launches)} + total_ms"""
        ms = (ctypes.c_float * 6)()
        n = (ctypes.c_uint32 * 5)()
        if self._L.h264bsdmiReplayTimings(self._h, ms, n) != 0:
            raise RuntimeError("h264bsdmiReplayTimings failed")
        tm = getattr(self, "_timed_mask", 31)
        out = {k: (float(ms[i]) if (tm >> i) & 1 else None, int(n[i])) for i, k in enumerate(self.KERNELS)}
        out["total_ms"] = float(ms[5])
        return out
