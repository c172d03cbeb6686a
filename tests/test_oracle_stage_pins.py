"""The oracle's intra-prediction and deblocking stages against the compiled reference's own functions.

tests/test_oracle_vs_ref.py pins the transform, inter prediction and colour conversion of oracle/pixel_oracle.c to
h264bsdProcessBlock / h264bsdPredictSamples / h264bsdConvertTo*; intra prediction and the in-loop filter were pinned only
end to end, through whole streams.  Here the reference's functions are driven directly (ctypes, oracle/_ref):

  * h264bsdIntra16x16Prediction / h264bsdIntra4x4Prediction / h264bsdIntraChromaPrediction
    (src/h264bsd_intra_prediction.c:627, 701, 845) on one macroblock with neighbours of known content, every prediction
    mode that the availability pattern allows, every availability pattern — against oracle_recon() on a 3 x 2 macroblock
    frame job whose other macroblocks are I_PCM;
  * h264bsdFilterPicture (src/h264bsd_deblocking.c:575) on random pictures with random macroblock types (intra, P_Skip,
    16x16, 16x8, 8x16, 8x8), coefficient maps, motion, references, QP 0..51, filter offsets -12..12, chroma QP offsets and
    per-macroblock edge flags — against oracle_deblock().

CPU only."""
import ctypes
import struct

import numpy as np
import pytest

from oracle import pyoracle
import h264bsd_amd
from jobgen import QPC, Z_X, Z_Y, _i4_modes, build_job


class MvT(ctypes.Structure):             # reference src/h264bsd_macroblock_layer.h:117-122
    _fields_ = [("hor", ctypes.c_int16), ("ver", ctypes.c_int16)]


class MbStorage(ctypes.Structure):       # reference src/h264bsd_macroblock_layer.h:162-185 (generic C branch: i16 totalCoeff)
    pass


MbStorage._fields_ = [("mbType", ctypes.c_int), ("sliceId", ctypes.c_uint32), ("disableDeblockingFilterIdc", ctypes.c_uint32),
                      ("filterOffsetA", ctypes.c_int32), ("filterOffsetB", ctypes.c_int32), ("qpY", ctypes.c_uint32),
                      ("chromaQpIndexOffset", ctypes.c_int32), ("totalCoeff", ctypes.c_int16 * 27), ("intra4x4PredMode", ctypes.c_uint8 * 16),
                      ("refPic", ctypes.c_uint32 * 4), ("refAddr", ctypes.c_void_p * 4), ("mv", MvT * 16), ("decoded", ctypes.c_uint32),
                      ("mbA", ctypes.POINTER(MbStorage)), ("mbB", ctypes.POINTER(MbStorage)), ("mbC", ctypes.POINTER(MbStorage)),
                      ("mbD", ctypes.POINTER(MbStorage))]


class MbPred(ctypes.Structure):          # :124-131
    _fields_ = [("prevIntra4x4PredModeFlag", ctypes.c_uint32 * 16), ("remIntra4x4PredMode", ctypes.c_uint32 * 16),
                ("intraChromaPredMode", ctypes.c_uint32), ("refIdxL0", ctypes.c_uint32 * 4), ("mvdL0", MvT * 4)]


class SubMbPred(ctypes.Structure):       # :133-138
    _fields_ = [("subMbType", ctypes.c_int * 4), ("refIdxL0", ctypes.c_uint32 * 4), ("mvdL0", (MvT * 4) * 4)]


class Residual(ctypes.Structure):        # :140-150
    _fields_ = [("totalCoeff", ctypes.c_int16 * 27), ("level", (ctypes.c_int32 * 16) * 26), ("coeffMap", ctypes.c_uint32 * 24)]


class MacroblockLayer(ctypes.Structure):  # :152-160
    _fields_ = [("mbType", ctypes.c_int), ("codedBlockPattern", ctypes.c_uint32), ("mbQpDelta", ctypes.c_int32), ("mbPred", MbPred),
                ("subMbPred", SubMbPred), ("residual", Residual)]


class ImageT(ctypes.Structure):          # reference src/h264bsd_image.h:46-55
    _fields_ = [("data", ctypes.c_void_p), ("width", ctypes.c_uint32), ("height", ctypes.c_uint32),
                ("luma", ctypes.c_void_p), ("cb", ctypes.c_void_p), ("cr", ctypes.c_void_p)]


P_SKIP, P_16x16, P_16x8, P_8x16, P_8x8, I_4x4, I_16x16_BASE, I_PCM = 0, 1, 2, 3, 4, 6, 7, 31   # mbType_e, :52-84


@pytest.fixture(scope="module")
def ref():
    pyoracle.build(ref=True)
    lib = ctypes.CDLL(pyoracle.REF_SO)
    lib.h264bsdIntra16x16Prediction.restype = ctypes.c_uint32
    lib.h264bsdIntra4x4Prediction.restype = ctypes.c_uint32
    lib.h264bsdIntraChromaPrediction.restype = ctypes.c_uint32
    lib.h264bsdFilterPicture.restype = None
    return lib


# ------------------------------------------------------------------ intra prediction
def _intra_blob(lib, rng, kind, avail, l16_mode, chroma_mode, i4modes):
    """3 x 2 macroblocks; the one under test at (1,1): neighbours D (0,0), B (1,0), C (2,0), A (0,1) are I_PCM with random
    samples, no coefficients anywhere -> the picture after oracle_recon() holds the bare prediction."""
    wmb, hmb, n = 3, 2, 6
    rec_off, mv_off = 128, 128 + n * 32
    coef_off = mv_off + n * 64
    cap = coef_off + (n * 27 + 2) * 32 + 8192
    buf = np.zeros(cap, dtype=np.uint8)
    recs = buf[rec_off:rec_off + n * 32].reshape(n, 32)
    nblk = 0
    for a in range(n):
        r = recs[a]
        struct.pack_into("<I", r, 12, nblk)
        if a == 4:
            r[0] = kind; r[1] = 26; r[2] = QPC[26]; r[3] = avail
            r[4] = (l16_mode if kind == 2 else 0) | (chroma_mode << 2)
            if kind == 1:
                for z in range(16): r[24 + (z >> 1)] |= i4modes[z] << ((z & 1) * 4)
        else:
            r[0] = 3                                                   # I_PCM
            buf[coef_off + 32 * nblk: coef_off + 32 * nblk + 384] = rng.integers(0, 256, 384, dtype=np.uint8)
            nblk += 12
    struct.pack_into("<IIHHIBBBBIII", buf, 0, 0x314A4648, 0, wmb, hmb, n, 0, 1, 1, 0, rec_off, mv_off, 0)
    struct.pack_into("<I", buf, 36, coef_off)
    assert lib.h264bsdmiJobFinalize(ctypes.c_void_p(buf.ctypes.data), cap, nblk) == 0
    return bytes(buf[:struct.unpack_from("<I", buf, 4)[0]])


def _neighbour_pels(frame, wmb, hmb, mbx, mby):
    """pelAbove / pelLeft as h264bsdGetNeighbourPels lays them out (src/h264bsd_intra_prediction.c:478-497, 541-625)"""
    W, H = wmb * 16, hmb * 16
    Y = frame[:W * H].reshape(H, W)
    C = frame[W * H:].reshape(2, H // 2, W // 2)
    x, y = mbx * 16, mby * 16
    above = np.zeros(1 + 16 + 4 + 1 + 8 + 1 + 8, dtype=np.uint8)
    above[0] = Y[y - 1, x - 1]; above[1:17] = Y[y - 1, x:x + 16]; above[17:21] = Y[y - 1, x + 16:x + 20]
    cx, cy = mbx * 8, mby * 8
    for p in range(2):
        above[21 + 9 * p] = C[p, cy - 1, cx - 1]; above[22 + 9 * p:30 + 9 * p] = C[p, cy - 1, cx:cx + 8]
    left = np.zeros(32, dtype=np.uint8)
    left[:16] = Y[y:y + 16, x - 1]
    for p in range(2): left[16 + 8 * p:24 + 8 * p] = C[p, cy:cy + 8, cx - 1]
    return above, left


def _mb_from_frame(frame, wmb, hmb, mbx, mby):
    W, H = wmb * 16, hmb * 16
    Y = frame[:W * H].reshape(H, W)
    C = frame[W * H:].reshape(2, H // 2, W // 2)
    return np.concatenate([Y[mby * 16:mby * 16 + 16, mbx * 16:mbx * 16 + 16].ravel(), C[0, mby * 8:mby * 8 + 8, mbx * 8:mbx * 8 + 8].ravel(),
                           C[1, mby * 8:mby * 8 + 8, mbx * 8:mbx * 8 + 8].ravel()])


def _prev_rem(modes_z, avail):
    """prev_intra4x4_pred_mode_flag / rem_intra4x4_pred_mode that make the reference derive modes_z (8.3.1.1; the neighbouring
    macroblocks are I_PCM: their blocks count as DC when available, and an unavailable neighbour makes the prediction DC)"""
    A, B = avail & 1, avail & 2
    grid = np.zeros((4, 4), int)
    for z in range(16): grid[Z_Y[z], Z_X[z]] = modes_z[z]
    prev, rem = [0] * 16, [0] * 16
    for z in range(16):
        bx, by = Z_X[z], Z_Y[z]
        ma = grid[by, bx - 1] if bx > 0 else (2 if A else None)
        mb = grid[by - 1, bx] if by > 0 else (2 if B else None)
        pred = 2 if ma is None or mb is None else min(ma, mb)
        m = modes_z[z]
        if m == pred: prev[z] = 1
        else: rem[z] = m if m < pred else m - 1
    return prev, rem


@pytest.mark.parametrize("avail", range(16))
def test_intra_prediction_matches_the_reference_functions(ref, built, avail):
    """every availability pattern of (A, B, C, D) x every Intra16x16 / chroma / Intra4x4 mode it allows"""
    orc = pyoracle.oracle_lib()
    lib = built.lib()
    rng = np.random.default_rng(1000 + avail)
    wmb, hmb = 3, 2
    cases = []
    l16 = [2] + ([0] if avail & 2 else []) + ([1] if avail & 1 else []) + ([3] if (avail & 11) == 11 else [])
    cm = [0] + ([1] if avail & 1 else []) + ([2] if avail & 2 else []) + ([3] if (avail & 11) == 11 else [])
    for m in l16:
        for c in cm: cases.append((2, m, c, None))
    for _ in range(24): cases.append((1, 0, int(rng.choice(cm)), _i4_modes(rng, avail)))
    seen_modes = set()
    for kind, m16, cmode, i4 in cases:
        blob = _intra_blob(lib, rng, kind, avail, m16, cmode, i4)
        dpb = pyoracle.OracleDpb(blob)
        frame = dpb.decode(blob, deblock=False).copy()
        want = _mb_from_frame(frame, wmb, hmb, 1, 1)
        above, left = _neighbour_pels(frame, wmb, hmb, 1, 1)
        mbs = (MbStorage * 6)()
        for a in range(6): mbs[a].mbType = I_PCM; mbs[a].sliceId = 0
        cur = mbs[4]
        cur.mbType = I_4x4 if kind == 1 else I_16x16_BASE + m16
        if avail & 1: cur.mbA = ctypes.pointer(mbs[3])
        if avail & 2: cur.mbB = ctypes.pointer(mbs[1])
        if avail & 4: cur.mbC = ctypes.pointer(mbs[2])
        if avail & 8: cur.mbD = ctypes.pointer(mbs[0])
        layer = MacroblockLayer()
        layer.mbType = cur.mbType
        layer.mbPred.intraChromaPredMode = cmode
        data = np.zeros(384 + 64, dtype=np.uint8)
        if kind == 1:
            prev, rem = _prev_rem(i4, avail)
            for z in range(16):
                layer.mbPred.prevIntra4x4PredModeFlag[z] = prev[z]; layer.mbPred.remIntra4x4PredMode[z] = rem[z]
                seen_modes.add(i4[z])
            rc = ref.h264bsdIntra4x4Prediction(ctypes.byref(cur), ctypes.c_void_p(data.ctypes.data), ctypes.byref(layer),
                                               ctypes.c_void_p(above.ctypes.data), ctypes.c_void_p(left.ctypes.data), 0)
            assert rc == 0
            assert [cur.intra4x4PredMode[z] for z in range(16)] == i4, "the reference derived other modes than the job carries"
        else:
            rc = ref.h264bsdIntra16x16Prediction(ctypes.byref(cur), ctypes.c_void_p(data.ctypes.data), ctypes.byref(layer.residual.level),
                                                 ctypes.c_void_p(above.ctypes.data), ctypes.c_void_p(left.ctypes.data), 0)
            assert rc == 0
        level16 = ctypes.cast(ctypes.byref(layer.residual.level, 16 * 16 * 4), ctypes.c_void_p)
        rc = ref.h264bsdIntraChromaPrediction(ctypes.byref(cur), ctypes.c_void_p(data.ctypes.data + 256), level16,
                                              ctypes.c_void_p(above.ctypes.data + 21), ctypes.c_void_p(left.ctypes.data + 16), cmode, 0)
        assert rc == 0
        got = data[:384]
        assert np.array_equal(got, want), f"kind {kind} mode {m16} chroma {cmode} i4 {i4} avail {avail}: {np.count_nonzero(got != want)} samples differ"
    if avail == 15: assert seen_modes == set(range(9))


# ------------------------------------------------------------------ deblocking
def _patch_partitions(recs, mvs16, rng):
    """give every inter macroblock of a jobgen picture (records and dense vectors, before the job is finished) a macroblock
    type (Skip / 16x16 / 16x8 / 8x16 / 8x8) with motion and references to match, and the FJ_PARTS_* hint the parser would
    set; returns the types"""
    n = recs.shape[0]
    mvs = mvs16.reshape(n, 4, 4, 2)                                   # [mb][by][bx][xy], raster
    types = []
    for a in range(n):
        if recs[a, 0] != 0:
            types.append(I_4x4 if recs[a, 0] in (1, 3) else I_16x16_BASE)
            continue
        t = int(rng.choice([P_SKIP, P_16x16, P_16x8, P_8x16, P_8x8]))
        base = rng.integers(-40, 41, 2)
        small = lambda: base + rng.integers(-6, 7, 2)                 # differences around the threshold of 4 quarter samples
        refs = recs[a, 16:20].copy()
        if t in (P_SKIP, P_16x16):
            mvs[a, :, :] = small(); refs[:] = refs[0]; parts = 1
        elif t == P_16x8:
            mvs[a, :2] = small(); mvs[a, 2:] = small(); refs[1] = refs[0]; refs[3] = refs[2]; parts = 2
        elif t == P_8x16:
            mvs[a, :, :2] = small(); mvs[a, :, 2:] = small(); refs[2] = refs[0]; refs[3] = refs[1]; parts = 3
        else:
            for by in range(4):
                for bx in range(4): mvs[a, by, bx] = small()
            parts = 0
        recs[a, 16:20] = refs
        recs[a, 4] = (int(recs[a, 4]) & 0x8F) | (parts << 4)
        types.append(t)
    return types


@pytest.mark.parametrize("seed", range(24))
def test_deblocking_matches_h264bsdFilterPicture(ref, built, seed):
    orc = pyoracle.oracle_lib()
    rng = np.random.default_rng(5000 + seed)
    wmb, hmb = int(rng.integers(1, 7)), int(rng.integers(1, 6))
    n = wmb * hmb
    types = []
    blob = build_job(built.lib(), rng, wmb, hmb, 3, 4, [0, 1, 2], p_inter=0.75 if seed % 4 else 0.3, p_pcm=0.02,
                     patch=lambda recs_, mvs_: types.extend(_patch_partitions(recs_, mvs_, rng)))
    h = pyoracle.blob_header(blob)
    recs = np.frombuffer(blob, dtype=np.uint8, count=n * 32, offset=h["rec_off"]).reshape(n, 32)
    mvs = h264bsd_amd.job_mvs(blob)
    W, H = wmb * 16, hmb * 16
    start = rng.integers(0, 256, W * H * 3 // 2, dtype=np.uint8)
    if seed % 3 == 0:                                                 # smooth content: the filters switch on far more often
        start = (np.clip(rng.normal(128, 6, W * H * 3 // 2), 0, 255)).astype(np.uint8)
    ours = start.copy()
    buf = ctypes.create_string_buffer(blob, len(blob))
    assert orc.oracle_deblock(buf, ctypes.c_void_p(ours.ctypes.data)) == 0
    theirs = np.concatenate([start, np.zeros(64, dtype=np.uint8)])
    mbs = (MbStorage * n)()
    for a in range(n):
        r, m = recs[a], mbs[a]
        m.mbType = types[a]
        dbk = int(r[5])
        m.disableDeblockingFilterIdc = 0 if dbk else 1
        m.filterOffsetA = int(np.int8(r[6])); m.filterOffsetB = int(np.int8(r[7]))
        m.qpY = int(r[1]); m.chromaQpIndexOffset = int(np.int8(r[20]))
        coded = struct.unpack_from("<I", r, 8)[0]
        for z in range(16): m.totalCoeff[z] = 1 if (coded >> z) & 1 else 0
        for z in range(16):
            m.mv[z].hor = int(mvs[a, 4 * Z_Y[z] + Z_X[z], 0]); m.mv[z].ver = int(mvs[a, 4 * Z_Y[z] + Z_X[z], 1])
        for q in range(4): m.refAddr[q] = 0x1000 * (1 + int(r[16 + q]))
        if dbk & 1: m.mbA = ctypes.pointer(mbs[a - 1])
        if dbk & 2: m.mbB = ctypes.pointer(mbs[a - wmb])
    img = ImageT(theirs.ctypes.data, wmb, hmb, theirs.ctypes.data, theirs.ctypes.data + W * H, theirs.ctypes.data + W * H + W * H // 4)
    ref.h264bsdFilterPicture(ctypes.byref(img), mbs)
    diff = np.nonzero(ours != theirs[:ours.size])[0]
    assert diff.size == 0, f"{wmb}x{hmb}: {diff.size} samples differ, first at byte {int(diff[0])}"
    if seed % 3 == 0 and n >= 4: assert np.count_nonzero(ours != start) > 0, "the smooth pictures must actually get filtered"
