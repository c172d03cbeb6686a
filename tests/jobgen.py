"""Hand-built frame jobs with randomised content (test infrastructure).

Builds the packed per-picture work description of h264bsd_amd/csrc/framejob.h directly — no bitstream, no
parser — so that the kernels can be compared with the CPU oracle on inputs the bundled streams never produce:
all 16 luma / 64 chroma fractional positions, motion vectors far outside the picture, several reference
slots, I_PCM, every intra mode under every neighbour-availability pattern, arbitrary QPs / filter offsets /
per-MB deblocking flags.  The derived sections (schedules) are completed by the product's own
h264bsdmiJobFinalize(), i.e. by the same code the parser uses."""
import ctypes
import struct

import numpy as np

QPC = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 29,
       30, 31, 32, 32, 33, 34, 34, 35, 35, 36, 36, 37, 37, 37, 38, 38, 38, 39, 39, 39, 39]
Z_X = [0, 1, 0, 1, 2, 3, 2, 3, 0, 1, 0, 1, 2, 3, 2, 3]
Z_Y = [0, 0, 1, 1, 0, 0, 1, 1, 2, 2, 3, 3, 2, 2, 3, 3]


def z_of(x, y):
    return ((y >> 1) << 3) | ((x >> 1) << 2) | ((y & 1) << 1) | (x & 1)


def _i4_modes(rng, avail):
    """16 Intra4x4 modes (z order) valid for the MB-level availability bits"""
    A, B, C, D = avail & 1, avail & 2, avail & 4, avail & 8
    modes = []
    for z in range(16):
        bx, by = Z_X[z], Z_Y[z]
        left = bx > 0 or A
        top = by > 0 or B
        if bx > 0 and by > 0:
            tl = True
        elif by > 0:
            tl = bool(A)
        elif bx > 0:
            tl = bool(B)
        else:
            tl = bool(D)
        ok = [2]
        if top:
            ok += [0, 3, 7]
        if left:
            ok += [1, 8]
        if top and left and tl:
            ok += [4, 5, 6]
        modes.append(int(rng.choice(ok)))
    return modes


def _coef_block(rng, ac_only=False, density=0.3, amp=12):
    c = np.zeros(16, dtype=np.int16)
    m = rng.random(16) < density
    c[m] = rng.integers(-amp, amp + 1, int(m.sum()))
    if ac_only:
        c[0] = 0
    return c


LS2 = [16, 18, 20, 23, 25, 29]      # the largest level scale of qp % 6 (8.5.9), LS0 the scale of the chroma DC
LS0 = [10, 11, 13, 14, 16, 18]


def _to_bound(rng, coefs, first, n_luma, has_cdc, n_cac, qp_y, qp_c):
    """Rescale the levels of an inter macroblock so that the parser's magnitude bound (hd_resid.c: the largest block's sum of level magnitudes x
    the largest scale, <= 32735 per plane) lands within +-15 % of its limit: the macroblocks on either side of FJ_CODED_WIDE, and
    intermediates that use most of 16 bits on the packed side of it."""
    def rescale(lo, hi, scale, extra=0.0):
        blk = coefs[lo:hi].astype(np.int64)
        tot = np.abs(blk).reshape(-1, 16).sum(axis=1).max() if hi > lo else 0      # the bound is per block: the LARGEST block's sum
        if tot == 0:
            return
        target = rng.uniform(0.85, 1.15) * 32735.0 - extra
        f = max(0.0, target) / (tot * scale)
        new = np.clip(np.rint(blk * f), -2047, 2047).astype(np.int16)
        if np.abs(new).sum() == 0:
            new[np.argmax(np.abs(blk))] = 1
        coefs[lo:hi] = new
    o = 16 * first
    rescale(o, o + 16 * n_luma, LS2[qp_y % 6] << (qp_y // 6))
    o += 16 * n_luma
    dc_max = 0.0
    if has_cdc:
        q6 = qp_c // 6
        dc_max = float(np.abs(coefs[o:o + 8].astype(np.int64)).sum() * LS0[qp_c % 6] * (1 << (q6 - 1 if q6 >= 1 else 0)))
        o += 16
    rescale(o, o + 16 * n_cac, LS2[qp_c % 6] << (qp_c // 6), dc_max)


def build_job(lib, rng, wmb, hmb, cur_slot, n_slots, ref_slots, *, p_inter=0.6, p_pcm=0.03, mv_range=None, any_deblock=True, patch=None, near_bound=False):
    """One random picture.  ref_slots: slots holding valid pictures (empty -> intra only).
    patch(recs, mvs): called on the records [n][32] and the dense vectors [n][16][2] before the job is finished — a finished
    job carries its vectors in the records and the sparse section (framejob.h), the dense array here is only h264bsdmiJobFinalize's input."""
    n = wmb * hmb
    rec_off, mv_off = 128, 128 + n * 32
    coef_off = mv_off + n * 64
    cap = coef_off + (n * 27 + 2) * 32 + (n + 2) * 4 + n * 2 + n * 16 + n * 2 + n * 64 + 4096
    buf = np.zeros(cap, dtype=np.uint8)
    recs = buf[rec_off:rec_off + n * 32].reshape(n, 32)
    mvs = buf[mv_off:mv_off + n * 64].view(np.int16).reshape(n, 16, 2)
    coefs = buf[coef_off:].view(np.int16)
    nblk = 0
    mvr = mv_range if mv_range is not None else (wmb * 16 * 4 + 200)
    for a in range(n):
        x, y = a % wmb, a // wmb
        r = recs[a]
        qp = int(rng.integers(0, 52))
        cqp = int(rng.integers(-12, 13))
        r[1] = qp
        r[2] = QPC[min(51, max(0, qp + cqp))]
        r[20] = np.int8(cqp).view(np.uint8)
        r[6] = np.int8(2 * rng.integers(-6, 7)).view(np.uint8)
        r[7] = np.int8(2 * rng.integers(-6, 7)).view(np.uint8)
        dbk = 0
        if any_deblock and rng.random() < 0.9:
            dbk = 4 | (1 if x > 0 and rng.random() < 0.9 else 0) | (2 if y > 0 and rng.random() < 0.9 else 0)
        r[5] = dbk
        struct.pack_into("<I", r, 12, nblk)
        u = rng.random()
        coded = 0
        if ref_slots and u < p_inter:
            r[0] = 0                                                   # inter
            style = rng.random()
            if style < 0.35:                                           # uniform, often whole-sample
                mv = rng.integers(-mvr, mvr + 1, 2)
                if rng.random() < 0.5:
                    mv = (mv // 8) * 8
                mvs[a, :, :] = mv
                r[16:20] = rng.choice(ref_slots)
            else:                                                      # per-4x4 vectors, per-quadrant references
                base = rng.integers(-mvr, mvr + 1, 2)
                mvs[a] = base + rng.integers(-40, 41, (16, 2))
                r[16:20] = rng.choice(ref_slots, 4)
            if rng.random() < 0.5:
                for z in range(16):
                    if rng.random() < 0.3:
                        coefs[16 * nblk:16 * nblk + 16] = _coef_block(rng); nblk += 1; coded |= 1 << z
                if rng.random() < 0.5:
                    cdc = np.zeros(16, dtype=np.int16); cdc[:8] = rng.integers(-6, 7, 8)
                    coefs[16 * nblk:16 * nblk + 16] = cdc; nblk += 1; coded |= 1 << 25
                    for k in range(8):
                        if rng.random() < 0.3:
                            coefs[16 * nblk:16 * nblk + 16] = _coef_block(rng, ac_only=True); nblk += 1; coded |= 1 << (16 + k)
                if near_bound and coded:
                    first = struct.unpack_from("<I", r, 12)[0]
                    _to_bound(rng, coefs, first, bin(coded & 0xFFFF).count("1"), (coded >> 25) & 1, bin((coded >> 16) & 0xFF).count("1"), int(r[1]), int(r[2]))
        elif u < p_inter + p_pcm or (not ref_slots and u < p_pcm):
            r[0] = 3                                                   # I_PCM: 384 raw samples = 12 blocks
            r[1] = 0
            r[2] = QPC[min(51, max(0, cqp))]
            buf[coef_off + 32 * nblk: coef_off + 32 * nblk + 384] = rng.integers(0, 256, 384, dtype=np.uint8)
            nblk += 12
        else:
            avail = (1 if x > 0 else 0) | (2 if y > 0 else 0) | (4 if y > 0 and x + 1 < wmb else 0) | (8 if x > 0 and y > 0 else 0)
            if rng.random() < 0.3:
                avail &= int(rng.integers(0, 16))                      # pretend slice boundaries
            r[3] = avail
            cmodes = [0] + ([1] if avail & 1 else []) + ([2] if avail & 2 else []) + ([3] if (avail & 11) == 11 else [])
            chroma_mode = int(rng.choice(cmodes))
            if rng.random() < 0.55:
                r[0] = 1                                               # Intra4x4
                modes = _i4_modes(rng, avail)
                for z in range(16):
                    r[24 + (z >> 1)] |= modes[z] << ((z & 1) * 4)
                r[4] = chroma_mode << 2
                for z in range(16):
                    if rng.random() < 0.4:
                        coefs[16 * nblk:16 * nblk + 16] = _coef_block(rng); nblk += 1; coded |= 1 << z
            else:
                r[0] = 2                                               # Intra16x16
                lmodes = [2] + ([0] if avail & 2 else []) + ([1] if avail & 1 else []) + ([3] if (avail & 11) == 11 else [])
                r[4] = int(rng.choice(lmodes)) | (chroma_mode << 2)
                if rng.random() < 0.7:
                    coefs[16 * nblk:16 * nblk + 16] = rng.integers(-10, 11, 16); nblk += 1; coded |= 1 << 24
                if rng.random() < 0.5:
                    for z in range(16):
                        if rng.random() < 0.5:
                            coefs[16 * nblk:16 * nblk + 16] = _coef_block(rng, ac_only=True); nblk += 1; coded |= 1 << z
            if rng.random() < 0.6:
                cdc = np.zeros(16, dtype=np.int16); cdc[:8] = rng.integers(-6, 7, 8)
                coefs[16 * nblk:16 * nblk + 16] = cdc; nblk += 1; coded |= 1 << 25
                for k in range(8):
                    if rng.random() < 0.3:
                        coefs[16 * nblk:16 * nblk + 16] = _coef_block(rng, ac_only=True); nblk += 1; coded |= 1 << (16 + k)
        struct.pack_into("<I", r, 8, coded)
    if patch is not None:
        patch(recs, mvs)
    struct.pack_into("<IIHHIBBBBIII", buf, 0, 0x314A4648, 0, wmb, hmb, n, cur_slot, 0, n_slots, 0, rec_off, mv_off, 0)
    struct.pack_into("<I", buf, 36, coef_off)
    rc = lib.h264bsdmiJobFinalize(ctypes.c_void_p(buf.ctypes.data), cap, nblk)
    assert rc == 0
    total = struct.unpack_from("<I", buf, 4)[0]
    return bytes(buf[:total])
