"""The colour conversion of kernels/convert.hip.h runs in wrapping signed 16-bit halves (v_pk_*_i16): the identity it rests on,
checked for every (Y, Cb, Cr) against the reference's formula (src/h264bsd_decoder.c:1163-1370: integer BT.601, limited range)."""
import numpy as np


def _i16(x):
    return ((x + 32768) & 0xFFFF) - 32768


def test_packed_16_bit_form_equals_the_reference_formula_for_all_inputs():
    Y, Cb, Cr = np.meshgrid(np.arange(256, dtype=np.int64), np.arange(256, dtype=np.int64), np.arange(256, dtype=np.int64), indexing="ij")
    c, d, e = Y - 16, Cb - 128, Cr - 128
    clip = lambda v: np.clip(v, 0, 255)
    ref = (clip((298 * c + 409 * e + 128) >> 8), clip((298 * c - 100 * d - 208 * e + 128) >> 8), clip((298 * c + 516 * d + 128) >> 8))
    # conv_chroma(): per chroma sample
    tR = _i16(_i16(153 * Cr) - 20128)
    tG = _i16(_i16(-100 * Cb) + _i16(_i16(48 * Cr) + 6112))
    tB = _i16(_i16(4 * Cb) - 1056)
    oR, oG, oB = _i16(Cr - 144), _i16(112 - Cr), _i16(_i16(Cb + Cb) - 272)
    m = _i16(42 * Y)
    # conv_column(): shift, the two additions, v_sat_pk_u8_i16
    got = tuple(clip(_i16(_i16((_i16(m + t) >> 8) + o) + Y)) for t, o in ((tR, oR), (tG, oG), (tB, oB)))
    for r, g in zip(ref, got):
        assert np.array_equal(r, g)
    # every shifted sum fits a signed half BEFORE the shift (the claim in the header comment)
    for t in (153 * Cr - 20128, -100 * Cb + 48 * Cr + 6112, 4 * Cb - 1056):
        v = 42 * Y + t
        assert v.min() >= -32768 and v.max() <= 32767
