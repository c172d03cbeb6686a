"""Pins individual oracle stages against the REAL reference's own functions (oracle/_ref, binary built from
/root/reference/src) on random inputs, including the cases the bundled streams never reach (motion vectors far
outside the picture, every QP, every fractional position).  Skipped where the reference binary is absent."""
import ctypes
import os

import numpy as np
import pytest

from oracle import pyoracle

pytestmark = pytest.mark.skipif(not os.path.exists(pyoracle.REF_SO), reason="oracle/_ref not built (needs /root/reference)")


class ImageT(ctypes.Structure):          # reference src/h264bsd_image.h:46-55
    _fields_ = [("data", ctypes.c_void_p), ("width", ctypes.c_uint32), ("height", ctypes.c_uint32),
                ("luma", ctypes.c_void_p), ("cb", ctypes.c_void_p), ("cr", ctypes.c_void_p)]


class MvT(ctypes.Structure):             # reference src/h264bsd_macroblock_layer.h:118-122
    _fields_ = [("hor", ctypes.c_int16), ("ver", ctypes.c_int16)]


@pytest.fixture(scope="module")
def ref():
    lib = ctypes.CDLL(pyoracle.REF_SO)
    lib.h264bsdProcessBlock.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]
    lib.h264bsdProcessBlock.restype = ctypes.c_uint32
    lib.h264bsdProcessLumaDc.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
    lib.h264bsdProcessLumaDc.restype = None
    lib.h264bsdProcessChromaDc.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
    lib.h264bsdProcessChromaDc.restype = None
    return lib


def test_inverse_transform_matches_h264bsdProcessBlock(ref):
    rng = np.random.default_rng(3)
    orc = pyoracle.oracle_lib()
    zz = [0, 1, 4, 8, 5, 2, 3, 6, 9, 12, 13, 10, 7, 11, 14, 15]
    n_ok = 0
    for _ in range(4000):
        qp = int(rng.integers(0, 52))
        raster = np.zeros(16, dtype=np.int16)
        mask = rng.random(16) < rng.choice([0.1, 0.4, 1.0])
        raster[mask] = rng.integers(-30, 31, int(mask.sum()))
        scan = np.zeros(16, dtype=np.int32)              # the reference takes scan-order levels in place
        for k in range(16):
            scan[k] = raster[zz[k]]
        cmap = sum(1 << k for k in range(16) if scan[k])
        out = np.zeros(16, dtype=np.int32)
        orc.oracle_idct4x4(raster.ctypes.data, qp, 0, 0, out.ctypes.data)
        rc = ref.h264bsdProcessBlock(scan.ctypes.data, qp, 0, cmap)
        if rc != 0:                                        # residual outside [-512,511]: reference-only error path
            assert np.abs(out).max() > 511
            continue
        assert np.array_equal(out, scan)
        n_ok += 1
    assert n_ok > 2000


def test_dc_transforms_match_reference(ref):
    rng = np.random.default_rng(4)
    orc = pyoracle.oracle_lib()
    zz = [0, 1, 4, 8, 5, 2, 3, 6, 9, 12, 13, 10, 7, 11, 14, 15]
    for _ in range(2000):
        qp = int(rng.integers(0, 52))
        raster = rng.integers(-200, 201, 16).astype(np.int16)
        scan = np.array([raster[zz[k]] for k in range(16)], dtype=np.int32)
        out = np.zeros(16, dtype=np.int32)
        orc.oracle_luma_dc(raster.ctypes.data, qp, out.ctypes.data)
        ref.h264bsdProcessLumaDc(scan.ctypes.data, qp)
        assert np.array_equal(out, scan)
        c = rng.integers(-200, 201, 8).astype(np.int16)
        c32 = c.astype(np.int32)
        o2 = np.zeros(8, dtype=np.int32)
        orc.oracle_chroma_dc(c.ctypes.data, qp, o2.ctypes.data)
        orc.oracle_chroma_dc(c[4:].ctypes.data, qp, o2[4:].ctypes.data)
        ref.h264bsdProcessChromaDc(c32.ctypes.data, qp)
        assert np.array_equal(o2, c32)


def test_inter_prediction_matches_h264bsdPredictSamples(ref):
    """all partition sizes, all fractional positions, vectors up to far outside the picture"""
    rng = np.random.default_rng(5)
    orc = pyoracle.oracle_lib()
    wmb, hmb = 4, 3
    W, H = wmb * 16, hmb * 16
    frame = rng.integers(0, 256, W * H * 3 // 2, dtype=np.uint8)
    img = ImageT(frame.ctypes.data, wmb, hmb, None, None, None)
    ref.h264bsdPredictSamples.argtypes = [ctypes.c_void_p, ctypes.POINTER(MvT), ctypes.POINTER(ImageT)] + [ctypes.c_uint32] * 6
    ref.h264bsdPredictSamples.restype = None
    cb = frame[W * H:].ctypes.data
    cr = frame[W * H + W * H // 4:].ctypes.data
    sizes = [(16, 16), (16, 8), (8, 16), (8, 8), (8, 4), (4, 8), (4, 4)]
    for it in range(1500):
        pw, ph = sizes[it % 7]
        px = int(rng.integers(0, 16 // pw)) * pw
        py = int(rng.integers(0, 16 // ph)) * ph
        mbx, mby = int(rng.integers(0, wmb)), int(rng.integers(0, hmb))
        big = rng.random() < 0.3
        mvx = int(rng.integers(-8192, 8192)) if big else int(rng.integers(-90, 91))
        mvy = int(rng.integers(-2048, 2048)) if big else int(rng.integers(-90, 91))
        mv = MvT(mvx, mvy)
        data = np.zeros(384 + 64, dtype=np.uint8)
        base = data.ctypes.data
        off = (-base) % 16
        ref.h264bsdPredictSamples(base + off, ctypes.byref(mv), ctypes.byref(img), mbx * 16, mby * 16, px, py, pw, ph)
        got = data[off:off + 384]
        for y in range(ph):
            for x in range(pw):
                xa, ya = mbx * 16 + px + x, mby * 16 + py + y
                want = orc.oracle_luma_sample(frame.ctypes.data, W, H, xa + (mvx >> 2), ya + (mvy >> 2), mvx & 3, mvy & 3)
                assert got[16 * (py + y) + px + x] == want, (it, pw, ph, mvx, mvy, x, y)
        for y in range(ph // 2):
            for x in range(pw // 2):
                xc, yc = mbx * 8 + px // 2 + x, mby * 8 + py // 2 + y
                a = (xc + (mvx >> 3), yc + (mvy >> 3), mvx & 7, mvy & 7)
                assert got[256 + 8 * (py // 2 + y) + px // 2 + x] == orc.oracle_chroma_sample(cb, W // 2, H // 2, *a)
                assert got[320 + 8 * (py // 2 + y) + px // 2 + x] == orc.oracle_chroma_sample(cr, W // 2, H // 2, *a)


def test_colour_conversion_matches_reference_on_random_frames():
    rng = np.random.default_rng(6)
    r = pyoracle.RefDecoder()
    for (w, h) in ((16, 16), (48, 32), (640, 368)):
        yuv = rng.integers(0, 256, w * h * 3 // 2, dtype=np.uint8)
        for fmt in range(3):
            assert np.array_equal(pyoracle.oracle_convert(fmt, w, h, yuv), r.convert(fmt, w, h, yuv))
