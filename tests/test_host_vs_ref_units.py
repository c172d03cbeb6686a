"""Unit-level differential tests of the host parser against the compiled reference (oracle/_ref), function by function:

* hd_cavlc_block  vs  h264bsdDecodeResidualBlockCavlc (src/h264bsd_cavlc.c:749-916) on random bit strings: same
  accept / reject decision, same total_coeff, same levels at the same positions, same number of bits consumed —
  including the damaged-stream case the reference accepts although the standard does not (a 15-coefficient block whose
  total_coeff + total_zeros is 16: the last level lands one element past the block; returned through *spill here);
* hd_residual_out_of_range  vs  h264bsdProcessBlock's range check (src/h264bsd_transform.c:97-234) on random blocks.

The host functions are internal to the product library (not exported), so the tests compile the host C files into a
scratch shared object of their own."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from oracle import pyoracle

CSRC = os.path.join(ROOT, "h264bsd_amd", "csrc")
ZIGZAG = [0, 1, 4, 8, 5, 2, 3, 6, 9, 12, 13, 10, 7, 11, 14, 15]


class BitReader(ctypes.Structure):
    _fields_ = [("buf", ctypes.c_void_p), ("size_bits", ctypes.c_uint32), ("pos", ctypes.c_uint32)]


class StrmData(ctypes.Structure):       # reference src/h264bsd_stream.h:46-53
    _fields_ = [("start", ctypes.c_void_p), ("cur", ctypes.c_void_p), ("bit_pos", ctypes.c_uint32),
                ("size", ctypes.c_uint32), ("read_bits", ctypes.c_uint32)]


@pytest.fixture(scope="module")
def libs(tmp_path_factory, built):
    if not os.path.exists(pyoracle.REF_SO):
        pytest.skip("oracle/_ref not built")
    so = str(tmp_path_factory.mktemp("hostdec") / "libhostdec_test.so")
    srcs = [os.path.join(CSRC, f) for f in ("hd_cavlc.c", "hd_resid.c")]
    subprocess.run(["gcc", "-O2", "-fPIC", "-shared", "-std=gnu11", "-I" + CSRC, "-DH264BSD_BUILD", *srcs, "-lpthread", "-o", so], check=True)
    hd, ref = ctypes.CDLL(so), ctypes.CDLL(pyoracle.REF_SO)
    hd.hd_cavlc_init()
    ref.h264bsdDecodeResidualBlockCavlc.restype = ctypes.c_uint32
    ref.h264bsdProcessBlock.restype = ctypes.c_uint32
    return hd, ref


def test_cavlc_block_accepts_and_decodes_exactly_like_the_reference(libs):
    hd, ref = libs
    rng = np.random.default_rng(2024)
    n_ok = n_spill = 0
    for it in range(60000):
        nbytes = int(rng.integers(1, 24))
        a, b, c = (rng.integers(0, 256, nbytes, dtype=np.uint8) for _ in range(3))
        raw = a if it % 3 == 0 else (a & b & c) if it % 3 == 1 else (a | b)      # dense, sparse (long prefixes), heavy
        buf = (ctypes.c_uint8 * (nbytes + 16))(*raw.tolist())
        start = int(rng.integers(0, 8))
        nc = int(rng.choice([-1, 0, 1, 2, 3, 4, 7, 8, 16]))
        mx = 4 if nc < 0 else int(rng.choice([15, 16]))
        br = BitReader(ctypes.addressof(buf), nbytes * 8, start)
        coef = (ctypes.c_int16 * 16)()
        spill = ctypes.c_int(0)
        r = hd.hd_cavlc_block(ctypes.byref(br), nc, mx, coef, ctypes.byref(spill))
        ours_ok = r >= 0 and br.pos <= br.size_bits
        sd = StrmData(ctypes.addressof(buf), ctypes.addressof(buf), start, nbytes, start)
        lvl = (ctypes.c_int32 * 40)()
        base = 8                                         # the reference is handed level[b] + 1 for 15-coefficient blocks
        rr = ref.h264bsdDecodeResidualBlockCavlc(ctypes.byref(sd), ctypes.byref(lvl, 4 * (base + (mx == 15))), nc, mx)
        ref_ok = (rr & 0xF) == 0
        assert ours_ok == ref_ok, (it, nc, mx, start, raw.tobytes().hex())
        if not ref_ok:
            continue
        expect = [0] * 16
        for i in range(16):
            if nc < 0:
                if i < 4:
                    expect[i] = lvl[base + i]
            else:
                expect[ZIGZAG[i]] = lvl[base + i]
        if mx == 15:
            expect[0] = 0
        assert r == (rr >> 4) & 0xFF and list(coef) == expect and br.pos == sd.read_bits and spill.value == lvl[base + 16], \
            (it, nc, mx, start, raw.tobytes().hex())
        n_ok += 1
        n_spill += lvl[base + 16] != 0
    assert n_ok > 25000 and n_spill > 300


def _to_scan(raster):
    out = [0] * 16
    for i in range(16):
        out[i] = raster[ZIGZAG[i]]
    return out


def test_residual_range_check_agrees_with_the_reference_transform(libs):
    hd, ref = libs
    rng = np.random.default_rng(7)
    n_bad = 0
    for it in range(40000):
        qp = int(rng.integers(0, 52))
        # levels around the size where the range limit is reached for this QP, few of them
        lim = max(2, 40000 // (10 << (qp // 6)))
        raster = np.zeros(16, dtype=np.int16)
        k = int(rng.integers(1, 5))
        pos = rng.choice(16, k, replace=False)
        raster[pos] = np.clip(rng.integers(-lim, lim + 1, k), -2500, 2500)
        blk = (ctypes.c_int16 * 16)(*raster.tolist())
        ours = hd.hd_residual_out_of_range(blk, 1, qp, qp, 0)                   # one luma block, z = 0, not Intra16x16
        scan = _to_scan(raster.tolist())                                          # the reference takes scan order + coeffMap
        data = (ctypes.c_int32 * 16)(*scan)
        cmap = sum(1 << i for i in range(16) if scan[i])
        theirs = ref.h264bsdProcessBlock(data, qp, 0, cmap) != 0
        assert bool(ours) == theirs, (it, qp, raster.tolist())
        n_bad += theirs
    assert 4000 < n_bad < 36000
