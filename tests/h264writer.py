"""A small H.264 baseline-profile bitstream WRITER (test infrastructure).

The reference ships three x264 streams that exercise one slice per picture, one slice group, POC type 2, no
I_PCM, no constrained intra prediction, no reference reordering, no MMCO and no encoder
is available offline.  This module writes syntactically valid random streams that DO use those features, so
that the host parser (and through it the kernels) can be compared with the compiled reference decoder
(oracle/_ref) on them.  It is a syntax generator, not an encoder: prediction modes, motion vector
differences and coefficient levels are random within what the standard (and the reference's error checks)
allow; the writer tracks exactly the state a decoder derives contexts from (neighbour availability per slice,
total_coeff for nC, Intra4x4PredMode for the mode predictor, which frames exist for ref_idx).

Written from the H.264 specification (7.3 syntax, 9.1 Exp-Golomb, 9.2 CAVLC tables 9-5 ... 9-10).
"""
import numpy as np

# ---------------------------------------------------------------- bits
class BitWriter:
    def __init__(self):
        self.bits = []

    def u(self, n, v):
        assert 0 <= v < (1 << n) if n else v == 0, (n, v)
        for i in range(n - 1, -1, -1):
            self.bits.append((v >> i) & 1)

    def ue(self, v):
        assert v >= 0
        v += 1
        n = v.bit_length()
        self.u(n - 1, 0)
        self.u(n, v)

    def se(self, v):
        self.ue(2 * v - 1 if v > 0 else -2 * v)

    def te(self, v, rng_max):            # rng_max = largest value
        if rng_max > 1:
            self.ue(v)
        else:
            self.u(1, 1 - v)

    def align_zero(self):
        self.bits.append(2)            # marker: resolved in bytes(), so that bit strings can be spliced before it

    def trailing(self):
        self.bits.append(1)
        self.bits.append(2)

    def bytes(self):
        out = []
        for b in self.bits:
            if b == 2:
                while len(out) % 8:
                    out.append(0)
            else:
                out.append(b)
        assert len(out) % 8 == 0
        return np.packbits(np.array(out, dtype=np.uint8)).tobytes()


def nal(nal_ref_idc, nal_type, rbsp, start_code=b"\x00\x00\x00\x01"):
    out = bytearray([(nal_ref_idc << 5) | nal_type])
    zeros = 0
    for b in rbsp:
        if zeros >= 2 and b <= 3:
            out.append(3)
            zeros = 0
        out.append(b)
        zeros = zeros + 1 if b == 0 else 0
    return start_code + bytes(out)


# ---------------------------------------------------------------- CAVLC tables (Tables 9-5, 9-7, 9-8, 9-9, 9-10)
CT_LEN = [
    [1, 0, 0, 0, 6, 2, 0, 0, 8, 6, 3, 0, 9, 8, 7, 5, 10, 9, 8, 6, 11, 10, 9, 7, 13, 11, 10, 8, 13, 13, 11, 9, 13, 13, 13, 10,
     14, 14, 13, 11, 14, 14, 14, 13, 15, 15, 14, 14, 15, 15, 15, 14, 16, 15, 15, 15, 16, 16, 16, 15, 16, 16, 16, 16, 16, 16, 16, 16],
    [2, 0, 0, 0, 6, 2, 0, 0, 6, 5, 3, 0, 7, 6, 6, 4, 8, 6, 6, 4, 8, 7, 7, 5, 9, 8, 8, 6, 11, 9, 9, 6, 11, 11, 11, 7,
     12, 11, 11, 9, 12, 12, 12, 11, 12, 12, 12, 11, 13, 13, 13, 12, 13, 13, 13, 13, 13, 14, 13, 13, 14, 14, 14, 13, 14, 14, 14, 14],
    [4, 0, 0, 0, 6, 4, 0, 0, 6, 5, 4, 0, 6, 5, 5, 4, 7, 5, 5, 4, 7, 5, 5, 4, 7, 6, 6, 4, 7, 6, 6, 4, 8, 7, 7, 5,
     8, 8, 7, 6, 9, 8, 8, 7, 9, 9, 8, 8, 9, 9, 9, 8, 10, 9, 9, 9, 10, 10, 10, 10, 10, 10, 10, 10, 10, 10, 10, 10],
]
CT_CODE = [
    [1, 0, 0, 0, 5, 1, 0, 0, 7, 4, 1, 0, 7, 6, 5, 3, 7, 6, 5, 3, 7, 6, 5, 4, 15, 6, 5, 4, 11, 14, 5, 4, 8, 10, 13, 4,
     15, 14, 9, 4, 11, 10, 13, 12, 15, 14, 9, 12, 11, 10, 13, 8, 15, 1, 9, 12, 11, 14, 13, 8, 7, 10, 9, 12, 4, 6, 5, 8],
    [3, 0, 0, 0, 11, 2, 0, 0, 7, 7, 3, 0, 7, 10, 9, 5, 7, 6, 5, 4, 4, 6, 5, 6, 7, 6, 5, 8, 15, 6, 5, 4, 11, 14, 13, 4,
     15, 10, 9, 4, 11, 14, 13, 12, 8, 10, 9, 8, 15, 14, 13, 12, 11, 10, 9, 12, 7, 11, 6, 8, 9, 8, 10, 1, 7, 6, 5, 4],
    [15, 0, 0, 0, 15, 14, 0, 0, 11, 15, 13, 0, 8, 12, 14, 12, 15, 10, 11, 11, 11, 8, 9, 10, 9, 14, 13, 9, 8, 10, 9, 8, 15, 14, 13, 13,
     11, 14, 10, 12, 15, 10, 13, 12, 11, 14, 9, 12, 8, 10, 13, 8, 13, 7, 9, 12, 9, 12, 11, 10, 5, 8, 7, 6, 1, 4, 3, 2],
]
CDC_LEN = [2, 0, 0, 0, 6, 1, 0, 0, 6, 6, 3, 0, 6, 7, 7, 6, 6, 8, 8, 7]
CDC_CODE = [1, 0, 0, 0, 7, 1, 0, 0, 4, 6, 1, 0, 3, 3, 2, 5, 2, 3, 2, 0]
TZ_LEN = [
    [1, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 9], [3, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 6, 6, 6, 6], [4, 3, 3, 3, 4, 4, 3, 3, 4, 5, 5, 6, 5, 6],
    [5, 3, 4, 4, 3, 3, 3, 4, 3, 4, 5, 5, 5], [4, 4, 4, 3, 3, 3, 3, 3, 4, 5, 4, 5], [6, 5, 3, 3, 3, 3, 3, 3, 4, 3, 6],
    [6, 5, 3, 3, 3, 2, 3, 4, 3, 6], [6, 4, 5, 3, 2, 2, 3, 3, 6], [6, 6, 4, 2, 2, 3, 2, 5], [5, 5, 3, 2, 2, 2, 4],
    [4, 4, 3, 3, 1, 3], [4, 4, 2, 1, 3], [3, 3, 1, 2], [2, 2, 1], [1, 1]]
TZ_CODE = [
    [1, 3, 2, 3, 2, 3, 2, 3, 2, 3, 2, 3, 2, 3, 2, 1], [7, 6, 5, 4, 3, 5, 4, 3, 2, 3, 2, 3, 2, 1, 0], [5, 7, 6, 5, 4, 3, 4, 3, 2, 3, 2, 1, 1, 0],
    [3, 7, 5, 4, 6, 5, 4, 3, 3, 2, 2, 1, 0], [5, 4, 3, 7, 6, 5, 4, 3, 2, 1, 1, 0], [1, 1, 7, 6, 5, 4, 3, 2, 1, 1, 0],
    [1, 1, 5, 4, 3, 3, 2, 1, 1, 0], [1, 1, 1, 3, 3, 2, 2, 1, 0], [1, 0, 1, 3, 2, 1, 1, 1], [1, 0, 1, 3, 2, 1, 1],
    [0, 1, 1, 2, 1, 3], [0, 1, 1, 1, 1], [0, 1, 1, 1], [0, 1, 1], [0, 1]]
CTZ_LEN = [[1, 2, 3, 3], [1, 2, 2], [1, 1]]
CTZ_CODE = [[1, 1, 1, 0], [1, 1, 0], [1, 0]]
RB_LEN = [[1, 1], [1, 2, 2], [2, 2, 2, 2], [2, 2, 2, 3, 3], [2, 2, 3, 3, 3, 3], [2, 3, 3, 3, 3, 3, 3], [3, 3, 3, 3, 3, 3, 3, 4, 5, 6, 7, 8, 9, 10, 11]]
RB_CODE = [[1, 0], [1, 1, 0], [3, 2, 1, 0], [3, 2, 1, 1, 0], [3, 2, 3, 2, 1, 0], [3, 0, 1, 3, 2, 5, 4], [7, 6, 5, 4, 3, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1]]


def write_residual_block(bw, coeffs, nc, max_coeff):
    """coeffs: list of max_coeff levels in SCAN order.  Returns total_coeff."""
    nz = [(i, c) for i, c in enumerate(coeffs) if c]
    total = len(nz)
    # trailing ones: up to 3 consecutive +-1 at the high-frequency end
    t1 = 0
    for _, c in reversed(nz):
        if abs(c) == 1 and t1 < 3:
            t1 += 1
        else:
            break
    if nc < 0:
        bw.u(CDC_LEN[4 * total + t1], CDC_CODE[4 * total + t1])
    elif nc >= 8:
        bw.u(6, 3 if total == 0 else ((total - 1) << 2) | t1)
    else:
        tab = 0 if nc < 2 else 1 if nc < 4 else 2
        bw.u(CT_LEN[tab][4 * total + t1], CT_CODE[tab][4 * total + t1])
    if total == 0:
        return 0
    levels = [c for _, c in reversed(nz)]          # highest frequency first
    for c in levels[:t1]:
        bw.u(1, 1 if c < 0 else 0)
    suffix_len = 1 if (total > 10 and t1 < 3) else 0
    for i, c in enumerate(levels[t1:]):
        code = 2 * abs(c) - (2 if c > 0 else 1)
        if i == 0 and t1 < 3:
            code -= 2
        if suffix_len == 0:
            if code < 14:
                bw.u(code, 0); bw.u(1, 1)
            elif code < 30:
                bw.u(14, 0); bw.u(1, 1); bw.u(4, code - 14)
            else:
                assert code - 30 < 4096
                bw.u(15, 0); bw.u(1, 1); bw.u(12, code - 30)
        else:
            prefix = code >> suffix_len
            if prefix < 15:
                bw.u(prefix, 0); bw.u(1, 1); bw.u(suffix_len, code & ((1 << suffix_len) - 1))
            else:
                assert code - (15 << suffix_len) < 4096
                bw.u(15, 0); bw.u(1, 1); bw.u(12, code - (15 << suffix_len))
        if suffix_len == 0:
            suffix_len = 1
        if abs(c) > (3 << (suffix_len - 1)) and suffix_len < 6:
            suffix_len += 1
    last = nz[-1][0]
    total_zeros = last + 1 - total
    if total < max_coeff:
        if nc < 0:
            bw.u(CTZ_LEN[total - 1][total_zeros], CTZ_CODE[total - 1][total_zeros])
        else:
            bw.u(TZ_LEN[total - 1][total_zeros], TZ_CODE[total - 1][total_zeros])
    zeros_left = total_zeros
    pos = [i for i, _ in nz]
    for k in range(total - 1, 0, -1):
        if zeros_left <= 0:
            break
        run = pos[k] - pos[k - 1] - 1
        t = min(zeros_left, 7) - 1
        bw.u(RB_LEN[t][run], RB_CODE[t][run])
        zeros_left -= run
    return total


# ---------------------------------------------------------------- parameter sets
def write_sps(p):
    bw = BitWriter()
    bw.u(8, 66); bw.u(8, 0xC0); bw.u(8, p.get("level", 40))
    bw.ue(p.get("sps_id", 0))
    bw.ue(p.get("log2_max_frame_num", 4) - 4)
    bw.ue(p["poc_type"])
    if p["poc_type"] == 0:
        bw.ue(p.get("log2_max_poc_lsb", 6) - 4)
    elif p["poc_type"] == 1:
        bw.u(1, p.get("delta_always_zero", 0))
        bw.se(p.get("offset_non_ref", 0)); bw.se(p.get("offset_top_bottom", 0))
        cyc = p.get("offsets_ref", [2])
        bw.ue(len(cyc))
        for o in cyc:
            bw.se(o)
    bw.ue(p["num_ref_frames"])
    bw.u(1, p.get("gaps", 0))
    bw.ue(p["wmb"] - 1); bw.ue(p["hmb"] - 1)
    bw.u(1, 1); bw.u(1, 1)                         # frame_mbs_only, direct_8x8_inference
    crop = p.get("crop")
    bw.u(1, 1 if crop else 0)
    if crop:
        for v in crop:
            bw.ue(v)
    reorder = p.get("num_reorder_frames")
    if reorder is None:
        bw.u(1, 0)
    else:                                          # minimal VUI with bitstream_restriction
        bw.u(1, 1)
        for _ in range(4):
            bw.u(1, 0)                             # aspect, overscan, video_signal, chroma_loc
        bw.u(1, 0); bw.u(1, 0); bw.u(1, 0)         # timing, nal_hrd, vcl_hrd
        bw.u(1, 0)                                 # pic_struct_present
        bw.u(1, 1)                                 # bitstream_restriction
        bw.u(1, 1); bw.ue(0); bw.ue(0); bw.ue(10); bw.ue(10)
        bw.ue(reorder); bw.ue(p.get("max_dec_frame_buffering", p["num_ref_frames"]))
    bw.trailing()
    return nal(3, 7, bw.bytes())


def write_pps(p, sps):
    bw = BitWriter()
    bw.ue(p.get("pps_id", 0)); bw.ue(sps.get("sps_id", 0))
    bw.u(1, 0)                                     # CAVLC
    bw.u(1, p.get("pic_order_present", 0))
    fmo = p.get("fmo")
    if not fmo:
        bw.ue(0)
    else:
        bw.ue(fmo["groups"] - 1)
        bw.ue(fmo["type"])
        if fmo["type"] == 0:
            for r in fmo["run_length"]:
                bw.ue(r - 1)
        elif fmo["type"] == 2:
            for tl, br in fmo["rects"]:
                bw.ue(tl); bw.ue(br)
        elif fmo["type"] in (3, 4, 5):
            bw.u(1, fmo["direction"]); bw.ue(fmo["rate"] - 1)
        elif fmo["type"] == 6:
            ids = fmo["ids"]
            bw.ue(len(ids) - 1)
            nb = 3 if fmo["groups"] > 4 else 2 if fmo["groups"] > 2 else 1
            for g in ids:
                bw.u(nb, g)
    bw.ue(p.get("num_ref_idx_active", 1) - 1); bw.ue(0)
    bw.u(1, 0); bw.u(2, 0)                         # weighted pred / bipred
    bw.se(p.get("pic_init_qp", 26) - 26); bw.se(0); bw.se(p.get("chroma_qp_offset", 0))
    bw.u(1, p.get("deblocking_control", 1)); bw.u(1, p.get("constrained_intra", 0)); bw.u(1, p.get("redundant_pic_cnt_present", 0))
    bw.trailing()
    return nal(3, 8, bw.bytes())


# ---------------------------------------------------------------- slice-group maps (8.2.2), writer side
def slice_group_map(fmo, wmb, hmb, change_cycle=0):
    n = wmb * hmb
    if not fmo:
        return [0] * n
    t, ng = fmo["type"], fmo["groups"]
    m = [0] * n
    if t == 0:
        i = 0
        while i < n:
            for g in range(ng):
                for j in range(fmo["run_length"][g]):
                    if i + j < n:
                        m[i + j] = g
                i += fmo["run_length"][g]
                if i >= n:
                    break
    elif t == 1:
        for i in range(n):
            m[i] = ((i % wmb) + (((i // wmb) * ng) // 2)) % ng
    elif t == 2:
        m = [ng - 1] * n
        for g in range(ng - 2, -1, -1):
            tl, br = fmo["rects"][g]
            for y in range(tl // wmb, br // wmb + 1):
                for x in range(tl % wmb, br % wmb + 1):
                    m[y * wmb + x] = g
    elif t in (4, 5):
        units0 = min(change_cycle * fmo["rate"], n)
        d = fmo["direction"]
        upper = n - units0 if d else units0
        if t == 4:
            m = [d if i < upper else 1 - d for i in range(n)]
        else:
            k = 0
            for x in range(wmb):
                for y in range(hmb):
                    m[y * wmb + x] = d if k < upper else 1 - d
                    k += 1
    elif t == 6:
        m = list(fmo["ids"])
    else:
        raise NotImplementedError("box-out map is not generated by the writer")
    return m


# ---------------------------------------------------------------- picture / slice / macroblock syntax
Z_X = [0, 1, 0, 1, 2, 3, 2, 3, 0, 1, 0, 1, 2, 3, 2, 3]
Z_Y = [0, 0, 1, 1, 0, 0, 1, 1, 2, 2, 3, 3, 2, 2, 3, 3]


def z_of(x, y):
    return ((y >> 1) << 3) | ((x >> 1) << 2) | ((y & 1) << 1) | (x & 1)


class MbState:
    __slots__ = ("kind", "slice_id", "tc", "i4")

    def __init__(self):
        self.kind = None          # 'I4', 'I16', 'PCM', 'P'
        self.slice_id = -1
        self.tc = [0] * 24
        self.i4 = [2] * 16


class StreamWriter:
    """Random baseline stream.  cfg keys: wmb, hmb, n_pics, seed, poc_type, num_ref_frames, num_ref_idx_active,
    slices_per_pic, idc (disable_deblocking_filter_idc choices), p_pcm, constrained_intra, fmo, idr_period,
    poc_pattern (display order offsets for poc_type 0), reorder (use ref_pic_list_reordering), mmco (use adaptive
    marking), chroma_qp_offset, multi_pps."""

    def __init__(self, **cfg):
        self.c = dict(wmb=4, hmb=3, n_pics=6, seed=1, poc_type=2, num_ref_frames=1, num_ref_idx_active=1, slices_per_pic=1,
                      idc=(0,), p_pcm=0.0, constrained_intra=0, fmo=None, idr_period=0, poc_pattern=None, reorder=False,
                      mmco=False, chroma_qp_offset=0, p_intra_in_p=0.2, p_skip=0.3, log2_max_frame_num=4,
                      num_reorder_frames=None, max_qp=28, aso=False, non_ref_every=0, gaps=0,
                      offset_non_ref=1, redundant=False, level=40, min_qp=6, huge_levels=False, overflow=0.0,
                      p_huge_mv=0.0)   # share of motion vector differences of up to +-2300 samples: far outside the picture, or out of range
        self.c.update(cfg)
        self.rng = np.random.default_rng(self.c["seed"])
        self.sps = dict(level=self.c["level"], poc_type=self.c["poc_type"], num_ref_frames=self.c["num_ref_frames"], wmb=self.c["wmb"], hmb=self.c["hmb"],
                        log2_max_frame_num=self.c["log2_max_frame_num"], gaps=self.c["gaps"],
                        offset_non_ref=self.c["offset_non_ref"], num_reorder_frames=self.c["num_reorder_frames"],
                        max_dec_frame_buffering=max(self.c["num_ref_frames"], 1) if self.c["num_reorder_frames"] is not None else None)
        self.pps = dict(num_ref_idx_active=self.c["num_ref_idx_active"], constrained_intra=self.c["constrained_intra"],
                        fmo=self.c["fmo"], chroma_qp_offset=self.c["chroma_qp_offset"], pic_init_qp=26,
                        pic_order_present=0, deblocking_control=1, redundant_pic_cnt_present=int(bool(self.c["redundant"])))

    # -- helpers
    def _avail(self, mbs, a, sid, dx, dy):
        wmb, hmb = self.c["wmb"], self.c["hmb"]
        x, y = a % wmb + dx, a // wmb + dy
        if x < 0 or y < 0 or x >= wmb or y >= hmb:
            return None
        m = mbs[y * wmb + x]
        return m if m.slice_id == sid else None

    def _nc(self, mbs, a, sid, cur, idx):
        """nC for luma block z=idx (0..15) or chroma AC (16..23)"""
        A = self._avail(mbs, a, sid, -1, 0)
        B = self._avail(mbs, a, sid, 0, -1)
        if idx < 16:
            x, y = Z_X[idx], Z_Y[idx]
            na = cur.tc[z_of(x - 1, y)] if x > 0 else (A.tc[z_of(3, y)] if A else None)
            nb = cur.tc[z_of(x, y - 1)] if y > 0 else (B.tc[z_of(x, 3)] if B else None)
        else:
            base = 16 if idx < 20 else 20
            k = idx - base
            x, y = k & 1, k >> 1
            na = cur.tc[base + k - 1] if x > 0 else (A.tc[base + 2 * y + 1] if A else None)
            nb = cur.tc[base + k - 2] if y > 0 else (B.tc[base + 2 + x] if B else None)
        if na is not None and nb is not None:
            return (na + nb + 1) >> 1
        return na if na is not None else nb if nb is not None else 0

    def _rand_block(self, n, density, amp=2, big=0.02):
        r = self.rng
        c = [0] * n
        if self.qp > 28:            # keep the reconstructed residual inside [-512, 511] (reference transform.c:184-188)
            amp, big = 1, 0.0
            density *= 0.5 if self.qp <= 40 else 0.15
        if self.c["overflow"] and self.qp >= 18 and r.random() < self.c["overflow"]:
            # one level around the size at which the dequantised coefficient alone leaves the residual range
            # [-512, 511]: the reference then fails the macroblock (transform.c:184-188) and conceals the slice.  Levels
            # are drawn on both sides of the limit so that near misses are exercised as well.
            limit = 32768 // (13 << (self.qp // 6))
            pos = int(r.integers(0, n))
            c[pos] = min(1500, max(1, int(limit * r.uniform(0.4, 1.6)))) * (1 if r.random() < 0.5 else -1)
            return c
        if self.c["huge_levels"] and self.qp <= 6 and r.random() < 0.5:
            # level_prefix 14 / 15 escapes and suffixLength growth (9.2.2.1): one huge level (+ a few small ones behind
            # it in scan order); at QP <= 6 a single coefficient of this size still reconstructs inside [-512, 511]
            pos = int(r.integers(0, n))
            c[pos] = int(r.integers(16, 900)) * (1 if r.random() < 0.5 else -1)
            for i in range(pos + 1, n):
                if r.random() < 0.15:
                    c[i] = 1 if r.random() < 0.5 else -1
            return c
        for i in range(n):
            if r.random() < density:
                v = int(r.integers(1, amp + 1)) * (1 if r.random() < 0.5 else -1)
                if r.random() < big:
                    v *= int(r.integers(3, 9))
                c[i] = v
        return c

    def _write_mb_intra(self, bw, mbs, a, sid, cur, is_p_slice, constrained, kind):
        r = self.rng
        A = self._avail(mbs, a, sid, -1, 0); B = self._avail(mbs, a, sid, 0, -1)
        C = self._avail(mbs, a, sid, 1, -1); D = self._avail(mbs, a, sid, -1, -1)

        def usable(m):
            return m is not None and not (constrained and m.kind == "P")
        av = (usable(A), usable(B), usable(C), usable(D))
        off = 5 if is_p_slice else 0
        cmodes = [0] + ([1] if av[0] else []) + ([2] if av[1] else []) + ([3] if av[0] and av[1] and av[3] else [])
        chroma_mode = int(r.choice(cmodes))
        cbp_chroma = int(r.choice([0, 1, 2]))
        if kind == "PCM":
            bw.ue(off + 25)
            bw.align_zero()
            for _ in range(384):
                bw.u(8, int(r.integers(0, 256)))
            cur.kind = "PCM"; cur.tc = [16] * 24
            return
        if kind == "I4":
            bw.ue(off + 0)
            modes = []
            for z in range(16):
                bx, by = Z_X[z], Z_Y[z]
                left = bx > 0 or av[0]
                top = by > 0 or av[1]
                tl = True if (bx > 0 and by > 0) else av[0] if by > 0 else av[1] if bx > 0 else av[3]
                ok = [2] + ([0, 3, 7] if top else []) + ([1, 8] if left else []) + ([4, 5, 6] if top and left and tl else [])
                mode = int(r.choice(ok))
                # predicted mode (8.3.1.1)
                def nmode(m, zz):
                    if m is None or (constrained and m.kind == "P"):
                        return -1
                    return m.i4[zz] if m.kind == "I4" else 2
                ma = modes[z_of(bx - 1, by)] if bx > 0 else nmode(A, z_of(3, by))
                mb_ = modes[z_of(bx, by - 1)] if by > 0 else nmode(B, z_of(bx, 3))
                pred = 2 if (ma < 0 or mb_ < 0) else min(ma, mb_)
                if mode == pred:
                    bw.u(1, 1)
                else:
                    bw.u(1, 0); bw.u(3, mode if mode < pred else mode - 1)
                modes.append(mode)
            cur.i4 = modes
            bw.ue(chroma_mode)
            cbp_luma = int(r.integers(0, 16))
            cbp = cbp_luma | (cbp_chroma << 4)
            inv_intra = {v: k for k, v in enumerate(CBP_INTRA)}
            bw.ue(inv_intra[cbp])
            cur.kind = "I4"
            is16 = False
        else:
            pmodes = [2] + ([0] if av[1] else []) + ([1] if av[0] else []) + ([3] if av[0] and av[1] and av[3] else [])
            pm = int(r.choice(pmodes))
            cbp_luma = 15 if r.random() < 0.4 else 0
            mbtype = 1 + pm + 4 * cbp_chroma + (12 if cbp_luma else 0)
            bw.ue(off + mbtype)
            bw.ue(chroma_mode)
            cbp = cbp_luma | (cbp_chroma << 4)
            cur.kind = "I16"
            is16 = True
        self._write_qp_and_residual(bw, mbs, a, sid, cur, cbp, is16)

    def _write_qp_and_residual(self, bw, mbs, a, sid, cur, cbp, is16):
        r = self.rng
        cur.tc = [0] * 24
        if not (cbp or is16):
            return
        lo, hi = max(-26, self.c["min_qp"] - self.qp), min(25, self.c["max_qp"] - self.qp)
        dq = int(r.integers(lo, hi + 1)) if r.random() < 0.3 else 0
        bw.se(dq)
        self.qp += dq
        if is16:
            write_residual_block(bw, self._rand_block(16, 0.3), self._nc(mbs, a, sid, cur, 0), 16)
        for z in range(16):
            if cbp & (1 << (z >> 2)):
                n = 15 if is16 else 16
                cur.tc[z] = write_residual_block(bw, self._rand_block(n, 0.25), self._nc(mbs, a, sid, cur, z), n)
        if cbp & 0x30:
            write_residual_block(bw, self._rand_block(4, 0.4), -1, 4)
            write_residual_block(bw, self._rand_block(4, 0.4), -1, 4)
        if cbp & 0x20:
            for k in range(8):
                cur.tc[16 + k] = write_residual_block(bw, self._rand_block(15, 0.2), self._nc(mbs, a, sid, cur, 16 + k), 15)

    def _write_mb_inter(self, bw, mbs, a, sid, cur, usable):
        """usable: ref_idx values this slice may use (entries of its final list that exist)"""
        r = self.rng
        n_active = self.cur_num_ref_idx
        ptype = int(r.choice([0, 0, 1, 2, 3, 3, 4])) if 0 in usable else int(r.choice([0, 1, 2, 3]))
        bw.ue(ptype)

        def mvd():
            if self.c["p_huge_mv"] and r.random() < self.c["p_huge_mv"]:        # (no draw when the option is off: streams of older fixtures stay as they are)
                return int(r.integers(-9200, 9201))
            return int(r.integers(-6, 7)) if r.random() < 0.7 else int(r.integers(-40, 41))
        if ptype <= 2:
            parts = 1 if ptype == 0 else 2
            if n_active > 1:
                for _ in range(parts):
                    bw.te(int(r.choice(usable)), n_active - 1)
            for _ in range(parts):
                bw.se(mvd()); bw.se(mvd())
        else:
            subs = [int(r.integers(0, 4)) for _ in range(4)]
            for s_ in subs:
                bw.ue(s_)
            if n_active > 1 and ptype != 4:
                for _ in range(4):
                    bw.te(int(r.choice(usable)), n_active - 1)
            for s_ in subs:
                for _ in range((1, 2, 2, 4)[s_]):
                    bw.se(mvd()); bw.se(mvd())
        cbp = int(r.choice([0, 0, int(r.integers(0, 48))]))
        inv_inter = {v: k for k, v in enumerate(CBP_INTER)}
        bw.ue(inv_inter[cbp])
        cur.kind = "P"
        self._write_qp_and_residual(bw, mbs, a, sid, cur, cbp, False)

    # -- the writer's model of reference marking (8.2.5) and list construction (8.2.4); entries are dicts
    #    {abs: running frame counter, exist: bool, lt: long-term index or None}
    def _init_list(self, refs, cur_abs, cur_fn, max_fn):
        short = sorted([f for f in refs if f["lt"] is None], key=lambda f: -f["abs"])
        long_ = sorted([f for f in refs if f["lt"] is not None], key=lambda f: f["lt"])
        return short + long_

    def _gen_reorder(self, bw, lst, refs, cur_abs, cur_fn, max_fn, n_active):
        """write ref_pic_list_reordering() with 1..3 random commands; returns the modified list"""
        r = self.rng
        lst = list(lst) + [None] * max(0, n_active + 1 - len(lst))
        pred = cur_fn
        n_cmd = int(r.integers(1, min(3, n_active) + 1))
        for k in range(n_cmd):
            cand = [f for f in refs if f["exist"]]        # a non-existing target is an error (reference dpb.c:288)
            f = cand[int(r.integers(0, len(cand)))]
            if f["lt"] is None:
                pic_num = cur_fn - (cur_abs - f["abs"])
                nowrap = pic_num if pic_num >= 0 else pic_num + max_fn
                if r.random() < 0.5:
                    d = (pred - nowrap) % max_fn or max_fn
                    bw.ue(0); bw.ue(d - 1)
                else:
                    d = (nowrap - pred) % max_fn or max_fn
                    bw.ue(1); bw.ue(d - 1)
                pred = nowrap
            else:
                bw.ue(2); bw.ue(f["lt"])
            # 8.2.4.3: insert at k, shift the rest, drop the later duplicate
            lst = lst[:k] + [f] + [x for x in lst[k:] if x is not f]
        bw.ue(3)
        return lst

    def build(self):
        c, r = self.c, self.rng
        wmb, hmb, n = c["wmb"], c["hmb"], c["wmb"] * c["hmb"]
        out = bytearray()
        out += write_sps(self.sps)
        out += write_pps(self.pps, self.sps)
        max_fn = 1 << c["log2_max_frame_num"]
        nrf = max(c["num_ref_frames"], 1)
        refs = []                  # reference frames as the decoder will hold them
        cur_abs = 0                # running frame counter; frame_num = cur_abs % max_fn
        max_lt = None              # MaxLongTermFrameIdx ("no long-term frame indices" = None)
        idr_id = 0
        since_idr = 0
        avoid_fn = None
        for pic in range(c["n_pics"]):
            is_idr = pic == 0 or bool(c["idr_period"] and pic % c["idr_period"] == 0)
            is_ref = is_idr or not (c["non_ref_every"] and pic % c["non_ref_every"] == c["non_ref_every"] - 1)
            if is_idr:
                refs = []; cur_abs = 0; since_idr = 0; max_lt = None
            elif c["gaps"] and r.random() < 0.25 and (len(refs) < nrf or any(f["lt"] is None for f in refs)) and avoid_fn is None:
                # skip 1..2 frame numbers: the decoder inserts "non-existing" frames (8.2.5.2).  (Not right after an
                # MMCO-5 picture: landing on its frame_num again would hide the access-unit boundary.)
                for _ in range(int(r.integers(1, 3))):
                    refs.append(dict(abs=cur_abs, exist=False, lt=None))
                    while len(refs) > nrf:
                        refs.remove(min([f for f in refs if f["lt"] is None], key=lambda f: f["abs"]))
                    cur_abs += 1
            avoid_fn = None
            cur_fn = cur_abs % max_fn
            can_p = not is_idr and any(f["exist"] for f in refs)
            is_p = can_p and r.random() < 0.85
            # ---- dec_ref_pic_marking of the picture (identical in all its slices)
            marking = None
            after = list(refs)
            cur_lt = None
            mmco5 = False
            if is_ref and is_idr:
                lt_flag = 1 if (c["mmco"] and r.random() < 0.3) else 0
                marking = ("idr", lt_flag)
            elif is_ref:
                ops = []
                fresh_idx = set()
                too_old = [f for f in after if f["lt"] is None and cur_abs - f["abs"] >= max_fn - 3]
                if (c["mmco"] and r.random() < 0.5) or too_old:
                    for f in too_old:
                        ops.append((1, cur_abs - f["abs"] - 1)); after.remove(f)
                    for _ in range(int(r.integers(0, 3))):
                        kind = int(r.choice([1, 2, 3, 3, 4, 6, 5] if c["mmco"] else [1]))
                        # the reference rejects more than one op 4 / 5 / 6 and ops 1-3 next to op 5 (slice_header.c DecRefPicMarking)
                        if (kind in (4, 5, 6) and any(o[0] == kind for o in ops)) or \
                                (kind == 5 and any(o[0] in (1, 2, 3, 6) for o in ops)) or (kind in (1, 2, 3) and mmco5):
                            continue
                        shorts = [f for f in after if f["lt"] is None]
                        longs = [f for f in after if f["lt"] is not None]
                        if kind == 1 and shorts:
                            f = shorts[int(r.integers(0, len(shorts)))]
                            ops.append((1, cur_abs - f["abs"] - 1)); after.remove(f)
                        elif kind == 2 and longs:
                            f = longs[int(r.integers(0, len(longs)))]
                            ops.append((2, f["lt"])); after.remove(f)
                        elif kind == 3 and shorts and max_lt is not None:
                            f = shorts[int(r.integers(0, len(shorts)))]
                            if not f["exist"]:
                                continue
                            idx = int(r.integers(0, max_lt + 1))
                            if idx in fresh_idx:          # do not reassign an index handed out by this picture
                                continue
                            fresh_idx.add(idx)
                            for g in [g for g in after if g["lt"] == idx and g is not f]:
                                after.remove(g)
                            ops.append((3, cur_abs - f["abs"] - 1, idx))
                            after[after.index(f)] = dict(f, lt=idx)
                        elif kind == 4:
                            v = int(r.integers(0, nrf + 1))
                            ops.append((4, v))
                            if cur_lt is not None and (v == 0 or cur_lt > v - 1):
                                ops.pop(); continue      # would orphan the index op 6 just gave the current picture
                            max_lt = v - 1 if v else None
                            after = [g for g in after if g["lt"] is None or (max_lt is not None and g["lt"] <= max_lt)]
                        elif kind == 5 and r.random() < 0.3 and cur_fn != 1:
                            # (the next picture has frame_num 1: with cur_fn == 1 nothing would mark the access-unit boundary)
                            ops.append((5,)); after = []; max_lt = None; mmco5 = True
                        elif kind == 6 and max_lt is not None and cur_lt is None:
                            idx = int(r.integers(0, max_lt + 1))
                            if idx in fresh_idx:
                                continue
                            fresh_idx.add(idx)
                            for g in [g for g in after if g["lt"] == idx]:
                                after.remove(g)
                            while len(after) >= nrf:      # op 6 needs a free frame when it executes
                                shorts = [f for f in after if f["lt"] is None]
                                f = min(shorts, key=lambda f: f["abs"]) if shorts else after[0]
                                ops.append((1, cur_abs - f["abs"] - 1) if shorts else (2, f["lt"]))
                                after.remove(f)
                            ops.append((6, idx)); cur_lt = idx
                    # make room for the current picture
                    while len(after) >= nrf:
                        shorts = [f for f in after if f["lt"] is None]
                        if shorts:
                            f = min(shorts, key=lambda f: f["abs"])
                            ops.append((1, cur_abs - f["abs"] - 1))
                        else:
                            f = after[0]
                            ops.append((2, f["lt"]))
                        after.remove(f)
                    marking = ("adaptive", ops)
                else:
                    marking = ("sliding",)
                    while len(after) >= nrf:
                        shorts = [f for f in after if f["lt"] is None]
                        if not shorts:
                            break
                        after.remove(min(shorts, key=lambda f: f["abs"]))
                    if len(after) >= nrf:      # only long-term frames left: sliding window cannot help
                        f = after[0]
                        marking = ("adaptive", [(2, f["lt"])]); after.remove(f)
            cycle = 0
            if c["fmo"] and c["fmo"]["type"] in (3, 4, 5):
                cycle = min((pic % 3) + 1, -(-n // c["fmo"]["rate"]))
            sgmap = slice_group_map(c["fmo"], wmb, hmb, change_cycle=cycle)
            groups = sorted(set(sgmap))
            slices = []
            for g in groups:
                members = [i for i in range(n) if sgmap[i] == g]
                k = min(c["slices_per_pic"], len(members))
                cuts = sorted(set([0] + r.choice(np.arange(1, len(members)), size=k - 1, replace=False).tolist())) if k > 1 else [0]
                for j, s0 in enumerate(cuts):
                    s1 = cuts[j + 1] if j + 1 < len(cuts) else len(members)
                    slices.append(members[s0:s1])
            if c["aso"]:
                slices = [slices[i] for i in r.permutation(len(slices)).tolist()]
            mbs = [MbState() for _ in range(n)]
            copies = []
            for sid, members in enumerate(slices):
                bw = BitWriter()
                bw.ue(members[0])
                slice_is_p = is_p and r.random() < 0.9
                bw.ue(int(r.choice([0, 5])) if slice_is_p else int(r.choice([2, 7])) if not is_p else 2)
                bw.ue(0)
                bw.u(c["log2_max_frame_num"], cur_fn)
                if is_idr:
                    bw.ue(idr_id)
                if c["poc_type"] == 0:
                    pat = c["poc_pattern"]
                    assert not pat or pat[0] == 0, "an IDR picture must have POC 0"
                    disp = since_idr if not pat else (since_idr // len(pat)) * len(pat) + pat[since_idr % len(pat)]
                    nb = self.sps.get("log2_max_poc_lsb", 6)
                    bw.u(nb, (2 * disp) % (1 << nb))
                elif c["poc_type"] == 1:
                    bw.se(0)
                head, bw = bw, BitWriter()        # redundant_pic_cnt goes between `head` and the rest
                usable = [0]
                if slice_is_p:
                    lst = self._init_list(refs, cur_abs, cur_fn, max_fn)
                    override = r.random() < 0.3
                    bw.u(1, 1 if override else 0)
                    if override:
                        self.cur_num_ref_idx = int(r.integers(1, min(len(lst), 4) + 1))
                        bw.ue(self.cur_num_ref_idx - 1)
                    else:
                        self.cur_num_ref_idx = c["num_ref_idx_active"]
                    want = c["reorder"] and r.random() < 0.6
                    if not any(f["exist"] for f in lst[: self.cur_num_ref_idx]):
                        want = True        # the initial list offers only non-existing frames: must reorder
                    bw.u(1, 1 if want else 0)
                    if want:
                        while True:
                            tmp = BitWriter()
                            new = self._gen_reorder(tmp, lst, refs, cur_abs, cur_fn, max_fn, self.cur_num_ref_idx)
                            if any(f is not None and f["exist"] for f in new[: self.cur_num_ref_idx]):
                                break
                        bw.bits += tmp.bits
                        lst = new
                    usable = [i for i, f in enumerate(lst[: self.cur_num_ref_idx]) if f is not None and f["exist"]]
                if is_ref:
                    if marking[0] == "idr":
                        bw.u(1, 0); bw.u(1, marking[1])
                    elif marking[0] == "sliding":
                        bw.u(1, 0)
                    else:
                        bw.u(1, 1)
                        for op in marking[1]:
                            bw.ue(op[0])
                            for v in op[1:]:
                                bw.ue(v)
                        bw.ue(0)
                self.qp = 26 + int(r.integers(-6, 3)) if not self.c["huge_levels"] else int(r.integers(0, 7))
                bw.se(self.qp - 26)
                idc = int(r.choice(c["idc"]))
                bw.ue(idc)
                if idc != 1:
                    bw.se(int(r.integers(-3, 4))); bw.se(int(r.integers(-3, 4)))
                if c["fmo"] and c["fmo"]["type"] in (3, 4, 5):
                    units = -(-n // c["fmo"]["rate"])
                    bw.u(int(np.ceil(np.log2(units + 1))), cycle)
                i = 0
                while i < len(members):
                    a = members[i]
                    cur = mbs[a]
                    cur.slice_id = sid
                    if slice_is_p:
                        run = 0
                        while 0 in usable and i + run < len(members) and r.random() < c["p_skip"]:
                            run += 1
                        bw.ue(run)
                        for k in range(run):
                            m = mbs[members[i + k]]
                            m.slice_id = sid; m.kind = "P"; m.tc = [0] * 24
                        i += run
                        if i >= len(members):
                            break
                        a = members[i]; cur = mbs[a]; cur.slice_id = sid
                    if slice_is_p and r.random() >= c["p_intra_in_p"]:
                        self._write_mb_inter(bw, mbs, a, sid, cur, usable)
                    else:
                        kind = "PCM" if r.random() < c["p_pcm"] else ("I4" if r.random() < 0.55 else "I16")
                        self._write_mb_intra(bw, mbs, a, sid, cur, slice_is_p, c["constrained_intra"], kind)
                    i += 1
                bw.trailing()

                def emit(rpc):
                    full = BitWriter()
                    full.bits = list(head.bits)
                    if c["redundant"]:
                        full.ue(rpc)
                    full.bits += bw.bits
                    return nal((1 + int(r.integers(0, 3))) if is_ref else 0, 5 if is_idr else 1, full.bytes())
                out += emit(0)
                if c["redundant"]:
                    # exact duplicates as redundant slices: before the picture is complete they are parsed again over
                    # already decoded macroblocks (reference slice_data.c:134,195), afterwards they are skipped
                    # (decoder.c:308)
                    copies.append(emit)
                    if r.random() < 0.6:
                        out += copies[int(r.integers(0, len(copies)))](int(r.integers(1, 3)))
            # ---- what the decoder holds after this picture
            if is_idr:
                idr_id = (idr_id + 1) % 16
            if is_ref:
                refs = after
                if marking[0] == "idr" and marking[1]:
                    refs.append(dict(abs=cur_abs, exist=True, lt=0)); max_lt = 0
                elif mmco5:
                    # the picture is inferred to have had frame_num 0 (7.4.3)
                    cur_abs = 0
                    refs.append(dict(abs=0, exist=True, lt=cur_lt))
                    since_idr = 0
                    avoid_fn = cur_fn
                else:
                    refs.append(dict(abs=cur_abs, exist=True, lt=cur_lt))
                cur_abs += 1
            since_idr += 1
        return bytes(out)


CBP_INTRA = [47, 31, 15, 0, 23, 27, 29, 30, 7, 11, 13, 14, 39, 43, 45, 46, 16, 3, 5, 10, 12, 19, 21, 26,
             28, 35, 37, 42, 44, 1, 2, 4, 8, 17, 18, 20, 24, 6, 9, 22, 25, 32, 33, 34, 36, 40, 38, 41]
CBP_INTER = [0, 16, 1, 2, 4, 8, 32, 3, 5, 10, 12, 15, 47, 7, 11, 13, 14, 6, 9, 31, 35, 37, 42, 44,
             33, 34, 36, 40, 39, 43, 45, 46, 17, 18, 20, 24, 19, 21, 26, 28, 23, 27, 29, 30, 22, 25, 38, 41]


def random_config(seed):
    """A random but valid combination of the writer's features (used by the synthetic-stream tests and sweeps)."""
    r = np.random.default_rng(seed + 1000)
    wmb, hmb = int(r.integers(2, 8)), int(r.integers(2, 7))
    n = wmb * hmb
    nrf = int(r.integers(1, 6))
    fmo = None
    t = int(r.integers(0, 10))
    if t == 0:
        g = int(r.integers(2, 5))
        fmo = dict(type=0, groups=g, run_length=[int(x) for x in r.integers(1, max(2, n // g + 1), size=g)])
    elif t == 1:
        fmo = dict(type=1, groups=int(r.integers(2, 6)))
    elif t == 2:
        a, b = int(r.integers(0, n // 2)), int(r.integers(n // 2, n))
        x0, y0, x1, y1 = a % wmb, a // wmb, b % wmb, b // wmb
        if x1 < x0:
            x0, x1 = x1, x0
        fmo = dict(type=2, groups=2, rects=[(y0 * wmb + x0, y1 * wmb + x1)])
    elif t == 3:
        fmo = dict(type=int(r.choice([4, 5])), groups=2, direction=int(r.integers(0, 2)), rate=int(r.integers(1, n)))
    elif t == 4:
        g = int(r.integers(2, 8))
        ids = [int(x) for x in r.integers(0, g, size=n)]
        for k in range(g):
            ids[k % n] = k
        fmo = dict(type=6, groups=g, ids=ids) if n >= g else None
    return dict(wmb=wmb, hmb=hmb, n_pics=int(r.integers(4, 25)), seed=seed, poc_type=int(r.integers(0, 3)),
                num_ref_frames=nrf, num_ref_idx_active=int(r.integers(1, nrf + 1)), slices_per_pic=int(r.integers(1, 4)),
                idc=tuple(r.choice([0, 1, 2], size=int(r.integers(1, 4))).tolist()), p_pcm=float(r.choice([0, 0, 0.1])),
                constrained_intra=int(r.integers(0, 2)), fmo=fmo, idr_period=int(r.choice([0, 0, 5, 9])),
                poc_pattern=[None, [0, 2, 1], [0, 3, 1, 2], [0, 4, 2, 1, 3]][int(r.integers(0, 4))],
                reorder=bool(r.integers(0, 2)), mmco=bool(r.integers(0, 2)), chroma_qp_offset=int(r.integers(-12, 13)),
                aso=bool(r.integers(0, 2)), non_ref_every=int(r.choice([0, 0, 3, 4])), gaps=int(r.integers(0, 2)),
                max_qp=int(r.choice([24, 28, 40, 51])), redundant=bool(r.integers(0, 4) == 0))
