#!/usr/bin/env python3
"""Parity sweep over the SEQUENCE PARAMETER SET: the SPS of a writer-made stream is replaced by one written field by
field with random contents — cropping rectangles, the whole VUI (aspect ratio incl. extended SAR, overscan, video signal
type, chroma location, timing, NAL / VCL HRD with several CPBs, bitstream restriction), levels, constraint flags — with
boundary and out-of-range values mixed in (the reference rejects e.g. max_bytes_per_pic_denom > 16, cpb_cnt > 32, a
cropping rectangle larger than the picture).  Compared with the compiled reference: the h264bsdDecode call trace, the
output pictures, and what the information calls return once headers are ready (h264bsdPicWidth / Height,
CroppingParams, VideoRange, MatrixCoefficients, SampleAspectRatio, Profile, CheckValidParamSets).  TEST TOOL (uses
oracle/).   usage: sweep_headers.py <first seed> <count> [pps|slice|nal|bytestream|multipps|multisps]      (multisps: three sequences under different SPS / PPS ids; multipps: several PPSs under different ids, slices pointed at them; bytestream: the Annex B framing is varied; nal: NAL units of the skipped types inserted in mid-stream; slice: 1-4 slice NAL units with random HEADERS are inserted between the stream's own; pps: the PICTURE parameter set is the random one: slice
group maps of all types with boundary values, QP offsets, reference counts, flags the baseline decoder rejects)"""
import sys, os, time, random, ctypes, hashlib
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import h264writer
from h264writer import BitWriter, nal
from oracle import pyoracle
from h264bsd_amd import capi

os.dup2(os.open(os.devnull, os.O_WRONLY), 2)
libc = ctypes.CDLL(None)


def pick(rng, usual, *odd):
    return usual if rng.random() < 0.8 or not odd else rng.choice(odd)


def hrd(bw, rng):
    cnt = pick(rng, rng.randrange(0, 4), 31, 32, 40)
    bw.ue(cnt); bw.u(4, rng.randrange(16)); bw.u(4, rng.randrange(16))
    for _ in range(min(cnt, 40) + 1):
        bw.ue(rng.choice([0, 1, 1000, 2 ** 20, 2 ** 32 - 2])); bw.ue(rng.choice([0, 5, 2 ** 16, 2 ** 32 - 2])); bw.u(1, rng.randrange(2))
    for _ in range(4):
        bw.u(5, rng.randrange(32))


def random_sps(cfg, rng):
    bw = BitWriter()
    bw.u(8, pick(rng, 66, 77, 100, 0)); bw.u(8, rng.choice([0xC0, 0xE0, 0x00, 0x40, 0xFF, 0x01])); bw.u(8, rng.choice([10, 11, 30, 40, 51, 9, 255]))
    bw.ue(pick(rng, 0, 31, 32))
    bw.ue(pick(rng, cfg.get("log2_max_frame_num", 4) - 4, 12, 13))
    bw.ue(cfg["poc_type"])
    if cfg["poc_type"] == 0:
        bw.ue(pick(rng, cfg.get("log2_max_poc_lsb", 6) - 4, 12, 13))
    elif cfg["poc_type"] == 1:
        bw.u(1, cfg.get("delta_always_zero", 0))
        bw.se(cfg.get("offset_non_ref", 0)); bw.se(cfg.get("offset_top_bottom", 0))
        cyc = cfg.get("offsets_ref", [2])
        bw.ue(len(cyc))
        for o in cyc:
            bw.se(o)
    bw.ue(pick(rng, cfg["num_ref_frames"], 16, 17))
    bw.u(1, cfg.get("gaps", 0))
    bw.ue(pick(rng, cfg["wmb"] - 1, cfg["wmb"], 0)); bw.ue(pick(rng, cfg["hmb"] - 1, cfg["hmb"] + 1))
    bw.u(1, pick(rng, 1, 0)); bw.u(1, rng.randrange(2))
    if rng.random() < 0.5:
        bw.u(1, 1)
        W, H = 8 * cfg["wmb"], 8 * cfg["hmb"]
        for lim in (W, W, H, H):
            bw.ue(pick(rng, rng.randrange(0, max(1, lim // 3)), lim, lim // 2, lim + 3))
    else:
        bw.u(1, 0)
    if rng.random() < 0.75:
        bw.u(1, 1)
        if rng.random() < 0.6:
            bw.u(1, 1); idc = rng.choice([0, 1, 2, 13, 14, 16, 17, 100, 255, 255]); bw.u(8, idc)
            if idc == 255:
                bw.u(16, rng.choice([0, 1, 40, 65535])); bw.u(16, rng.choice([0, 1, 33, 65535]))
        else:
            bw.u(1, 0)
        if rng.random() < 0.4:
            bw.u(1, 1); bw.u(1, rng.randrange(2))
        else:
            bw.u(1, 0)
        if rng.random() < 0.6:
            bw.u(1, 1); bw.u(3, rng.randrange(8)); bw.u(1, rng.randrange(2))
            if rng.random() < 0.6:
                bw.u(1, 1); bw.u(8, rng.randrange(256)); bw.u(8, rng.randrange(256)); bw.u(8, rng.choice([0, 1, 2, 5, 6, 7, 8, 200]))
            else:
                bw.u(1, 0)
        else:
            bw.u(1, 0)
        if rng.random() < 0.4:
            bw.u(1, 1); bw.ue(pick(rng, rng.randrange(6), 6, 100)); bw.ue(pick(rng, rng.randrange(6), 6))
        else:
            bw.u(1, 0)
        if rng.random() < 0.5:
            bw.u(1, 1); bw.u(32, rng.choice([0, 1, 1001, 2 ** 32 - 1])); bw.u(32, rng.choice([0, 1, 60000, 2 ** 32 - 1])); bw.u(1, rng.randrange(2))
        else:
            bw.u(1, 0)
        n_hrd = 0
        for _ in range(2):
            if rng.random() < 0.35:
                bw.u(1, 1); hrd(bw, rng); n_hrd += 1
            else:
                bw.u(1, 0)
        if n_hrd:
            bw.u(1, rng.randrange(2))
        bw.u(1, rng.randrange(2))
        if rng.random() < 0.6:
            bw.u(1, 1); bw.u(1, rng.randrange(2))
            bw.ue(pick(rng, rng.randrange(17), 17, 100)); bw.ue(pick(rng, rng.randrange(17), 17))
            bw.ue(pick(rng, rng.randrange(17), 17)); bw.ue(pick(rng, rng.randrange(17), 17))
            bw.ue(pick(rng, rng.randrange(0, 3), 17, 40)); bw.ue(pick(rng, cfg["num_ref_frames"], 0, 16, 17, 40))
        else:
            bw.u(1, 0)
    else:
        bw.u(1, 0)
    if rng.random() < 0.9:
        bw.trailing()
    else:
        bw.align_zero()
    return nal(3, 7, bw.bytes())


def random_pps(cfg, rng):
    """picture parameter set, every field random around what the stream needs (reference src/h264bsd_pic_param_set.c)"""
    n = cfg["wmb"] * cfg["hmb"]
    bw = BitWriter()
    bw.ue(pick(rng, 0, 1, 255, 256)); bw.ue(pick(rng, 0, 31, 32))
    bw.u(1, pick(rng, 0, 1))                                           # entropy_coding_mode: CABAC is rejected
    bw.u(1, pick(rng, cfg.get("pic_order_present", 0), 1))
    fmo = cfg.get("fmo")
    if rng.random() < 0.5:
        fmo = None if rng.random() < 0.4 else dict(type=rng.choice([0, 1, 2, 3, 4, 5, 6, 6, 7]), groups=rng.choice([2, 2, 3, 4, 5, 8, 9]))
    if not fmo:
        bw.ue(0)
    else:
        g, t = fmo["groups"], fmo["type"]
        bw.ue(g - 1); bw.ue(t)
        if t == 0:
            for r in fmo.get("run_length") or [pick(rng, rng.randrange(1, n + 1), n + 1, 2 ** 20) for _ in range(g)]:
                bw.ue(r - 1)
        elif t == 2:
            for tl, br in fmo.get("rects") or [(rng.randrange(n), pick(rng, rng.randrange(n), n, n + 5)) for _ in range(g - 1)]:
                bw.ue(tl); bw.ue(br)
        elif t in (3, 4, 5):
            bw.u(1, fmo.get("direction", rng.randrange(2))); bw.ue(fmo.get("rate", pick(rng, rng.randrange(1, n + 1), n + 1, n + 2)) - 1)
        elif t == 6:
            ids = fmo.get("ids") or [pick(rng, rng.randrange(g), g, 7) & 7 for _ in range(pick(rng, n, n - 1, n + 1, 1))]
            bw.ue(len(ids) - 1)
            nb = 3 if g > 4 else 2 if g > 2 else 1
            for x in ids:
                bw.u(nb, x & ((1 << nb) - 1))
    bw.ue(pick(rng, cfg.get("num_ref_idx_active", 1) - 1, 15, 31, 32)); bw.ue(pick(rng, 0, 31, 32))
    bw.u(1, pick(rng, 0, 1)); bw.u(2, pick(rng, 0, 1, 2, 3))
    bw.se(pick(rng, cfg.get("pic_init_qp", 26) - 26, -26, 25, -27, 26)); bw.se(pick(rng, 0, -26, 25, 26, -27))
    bw.se(pick(rng, cfg.get("chroma_qp_offset", 0), -12, 12, -13, 13))
    bw.u(1, pick(rng, cfg.get("deblocking_control", 1), 0)); bw.u(1, pick(rng, cfg.get("constrained_intra", 0), 1))
    bw.u(1, pick(rng, cfg.get("redundant_pic_cnt_present", 1 if cfg.get("redundant") else 0), 1, 0))
    if rng.random() < 0.15:                                            # more_rbsp_data(): the fields of the high profiles
        bw.u(1, rng.randrange(2)); bw.u(1, 0); bw.se(rng.randrange(-12, 13))
    if rng.random() < 0.9:
        bw.trailing()
    else:
        bw.align_zero()
    return nal(3, 8, bw.bytes())


def random_slice(cfg, rng):
    """a slice NAL unit whose HEADER is written field by field with random contents (reference src/h264bsd_slice_header.c:
    first_mb, type, parameter-set id, frame_num, idr_pic_id, POC fields, redundant_pic_cnt, reference counts, list
    reordering and reference marking command loops, QP delta, deblocking parameters, slice-group change cycle), followed
    by a few random bytes of "slice data" """
    n = cfg["wmb"] * cfg["hmb"]
    idr = rng.random() < 0.25
    ref_idc = rng.choice([0, 1, 2, 3]) if not idr else rng.choice([1, 2, 3, 3, 0])
    bw = BitWriter()
    bw.ue(pick(rng, rng.randrange(n), n, n + 7, 2 ** 20))
    st = pick(rng, rng.choice([0, 2, 5, 7]), 1, 3, 4, 6, 8, 9, 10)
    bw.ue(st)
    bw.ue(pick(rng, 0, 1, 255, 256))
    bw.u(cfg.get("log2_max_frame_num", 4), rng.randrange(1 << cfg.get("log2_max_frame_num", 4)))
    if idr:
        bw.ue(pick(rng, rng.randrange(16), 65535, 65536))
    if cfg["poc_type"] == 0:
        nb = cfg.get("log2_max_poc_lsb", 6)
        bw.u(nb, rng.randrange(1 << nb))
    elif cfg["poc_type"] == 1 and not cfg.get("delta_always_zero", 0):
        bw.se(pick(rng, rng.randrange(-4, 5), 2 ** 20, -2 ** 20))
    if cfg.get("redundant"):
        bw.ue(pick(rng, rng.randrange(3), 127, 128))
    if st % 5 == 0:
        if rng.random() < 0.5:
            bw.u(1, 1); bw.ue(pick(rng, rng.randrange(4), 15, 16, 31, 32))
        else:
            bw.u(1, 0)
        if rng.random() < 0.5:
            bw.u(1, 1)
            for _ in range(pick(rng, rng.randrange(4), 17, 34, 40)):
                op = pick(rng, rng.randrange(3), 4, 9)
                bw.ue(op); bw.ue(pick(rng, rng.randrange(8), 2 ** 16, 2 ** 20))
            if rng.random() < 0.9:
                bw.ue(3)
        else:
            bw.u(1, 0)
    if ref_idc:
        if idr:
            bw.u(1, rng.randrange(2)); bw.u(1, rng.randrange(2))
        elif rng.random() < 0.5:
            bw.u(1, 1)
            for _ in range(pick(rng, rng.randrange(4), 33, 36, 70)):
                op = pick(rng, rng.randrange(1, 7), 7, 12)
                bw.ue(op)
                if op in (1, 3):
                    bw.ue(pick(rng, rng.randrange(6), 2 ** 16))
                if op == 2:
                    bw.ue(pick(rng, rng.randrange(4), 40))
                if op in (3, 6):
                    bw.ue(pick(rng, rng.randrange(4), 16, 40))
                if op == 4:
                    bw.ue(pick(rng, rng.randrange(5), 16, 17, 40))
            if rng.random() < 0.9:
                bw.ue(0)
        else:
            bw.u(1, 0)
    bw.se(pick(rng, rng.randrange(-6, 7), 25, 26, -26, -27, 52, -52))
    if cfg.get("deblocking_control", 1):
        idc = pick(rng, rng.randrange(3), 3, 7)
        bw.ue(idc)
        if idc != 1:
            bw.se(pick(rng, rng.randrange(-6, 7), 7, -7)); bw.se(pick(rng, rng.randrange(-6, 7), 7, -7))
    if cfg.get("fmo") and cfg["fmo"]["type"] in (3, 4, 5):
        units = -(-n // cfg["fmo"]["rate"])
        nb = int(np.ceil(np.log2(units + 1)))
        bw.u(nb, rng.randrange(1 << nb))
    for _ in range(rng.randrange(0, 24)):
        bw.u(8, rng.randrange(256))
    if rng.random() < 0.7:
        bw.trailing()
    else:
        bw.align_zero()
    return nal(ref_idc, 5 if idr else 1, bw.bytes())


def _rbsp(nal_bytes):
    """payload of a NAL unit (after the one-byte header) without emulation-prevention bytes"""
    out = bytearray(); zeros = 0
    for b in nal_bytes:
        if zeros >= 2 and b == 3:
            zeros = 0
            continue
        out.append(b); zeros = zeros + 1 if b == 0 else 0
    return bytes(out)


def _read_ue(bits, pos):
    z = 0
    while bits[pos + z] == 0:
        z += 1
    v = 0
    for i in range(z + 1):
        v = (v << 1) | bits[pos + z + i]
    return v - 1, pos + 2 * z + 1


def with_pps_id(unit, new_id):
    """the slice NAL unit `unit` (start code + header + payload) with its pic_parameter_set_id replaced"""
    sc = 4 if unit[:4] == b"\x00\x00\x00\x01" else 3
    hdr = unit[sc]
    bits = np.unpackbits(np.frombuffer(_rbsp(unit[sc + 1:]), dtype=np.uint8)).tolist()
    _, p = _read_ue(bits, 0); _, p = _read_ue(bits, p)
    _, q = _read_ue(bits, p)
    bw = BitWriter(); bw.bits = bits[:p]; bw.ue(new_id)
    rest = bits[q:]
    last = len(rest) - 1 - rest[::-1].index(1)           # the stop bit: re-align behind it
    bw.bits += rest[:last + 1]; bw.align_zero()
    return nal((hdr >> 5) & 3, hdr & 31, bw.bytes(), start_code=unit[:sc])


def info_ref(lib, dec):
    dec = ctypes.c_void_p(dec)                      # (no argtypes are declared for the information calls: keep the pointer 64 bits wide)
    v = [ctypes.c_uint32() for _ in range(5)]
    lib.h264bsdCroppingParams(dec, *[ctypes.byref(x) for x in v])
    w, h = ctypes.c_uint32(), ctypes.c_uint32()
    lib.h264bsdSampleAspectRatio(dec, ctypes.byref(w), ctypes.byref(h))
    return (lib.h264bsdPicWidth(dec), lib.h264bsdPicHeight(dec), tuple(x.value for x in v), lib.h264bsdVideoRange(dec),
            lib.h264bsdMatrixCoefficients(dec), (w.value, h.value), lib.h264bsdProfile(dec), lib.h264bsdCheckValidParamSets(dec))


def info_ours(dec):
    return (dec.pic_width(), dec.pic_height(), dec.cropping_params(), dec.video_range(), dec.matrix_coefficients(),
            dec.sample_aspect_ratio(), dec.profile(), dec.check_valid_param_sets())


def run_ref(data):
    libc.mallopt(-6, 0xFF)
    try:
        lib = pyoracle.RefDecoder().lib
        buf = ctypes.create_string_buffer(data, len(data)); base = ctypes.addressof(buf)
        dec = lib.h264bsdAlloc(); lib.h264bsdInit(dec, 0)
        off = pid = stall = 0; rb = ctypes.c_uint32(0); trace, pics, infos = [], [], []
        a, b, c = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32()
        dims = [0, 0]
        def drain():
            # (a replaced active SPS makes h264bsdPicWidth return 0 while pictures of the old sequence still come out)
            if lib.h264bsdPicWidth(dec): dims[:] = [lib.h264bsdPicWidth(dec), lib.h264bsdPicHeight(dec)]
            w, h = dims
            while True:
                p = lib.h264bsdNextOutputPicture(dec, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
                if not p:
                    break
                pics.append((hashlib.sha1(ctypes.string_at(p, w * h * 384)).hexdigest(), a.value, b.value, c.value))
        while off < len(data):
            r = lib.h264bsdDecode(dec, base + off, len(data) - off, pid, ctypes.byref(rb)); trace.append((int(r), int(rb.value))); off += rb.value
            if r == 1:
                pid += 1; drain()
            elif r == 2:
                infos.append(info_ref(lib, dec))
            stall = stall + 1 if rb.value == 0 else 0
            if stall > 3:
                break
        infos.append(info_ref(lib, dec)) if any(t[0] == 2 for t in trace) else None
        lib.h264bsdFlushBuffer(dec); drain(); lib.h264bsdShutdown(dec); lib.h264bsdFree(dec)
        return trace, pics, infos
    finally:
        libc.mallopt(-6, 0)


def run_ours(data):
    pics, trace, infos, state = [], [], [], {"dpb": None}
    def on_job(blob):
        if state["dpb"] is None:
            state["dpb"] = pyoracle.OracleDpb(blob)
        state["dpb"].decode(blob)
    dec = capi.Decoder(0, capture=on_job)
    buf = ctypes.create_string_buffer(data, len(data)); base = ctypes.addressof(buf); off = pid = stall = 0
    def drain():
        while True:
            o = dec.next_output_info()
            if o is None:
                break
            slot, p, idr, nerr = o
            pics.append((hashlib.sha1(np.ascontiguousarray(state["dpb"].slots[slot][: state["dpb"].frame_bytes]).tobytes()).hexdigest(), p, idr, nerr))
    while off < len(data):
        r, rb = dec.decode(base + off, len(data) - off, pid); trace.append((r, rb)); off += rb
        if r == 1:
            pid += 1; drain()
        elif r == 2:
            state["dpb"] = None; infos.append(info_ours(dec))
        stall = stall + 1 if rb == 0 else 0
        if stall > 3:
            break
    infos.append(info_ours(dec)) if any(t[0] == 2 for t in trace) else None
    dec.flush_buffer(); drain(); dec.close()
    return trace, pics, infos


first, count = int(sys.argv[1]), int(sys.argv[2]); MODE = sys.argv[3] if len(sys.argv) > 3 else "sps"; bad = []; t0 = time.time(); n_ok = n_pics = 0
for seed in range(first, first + count):
    rng = random.Random(seed)
    cfg = h264writer.random_config(seed)
    cfg["n_pics"] = min(cfg.get("n_pics", 6), 4)
    data = h264writer.StreamWriter(**cfg).build()
    # the writer's stream starts with its SPS: find the second start code and splice
    second = data.index(b"\x00\x00\x00\x01", 4)
    assert data[4] & 31 == 7
    third = data.index(b"\x00\x00\x00\x01", second + 4)
    assert data[second + 4] & 31 == 8
    if MODE in ("pps", "slice", "nal", "bytestream", "multipps", "multisps"):
        if MODE == "pps":
            data = data[:second] + random_pps(cfg, rng) + data[third:]
            if rng.random() < 0.2:                     # ... and another one later
                cut = data.index(b"\x00\x00\x00\x01", len(data) // 2) if b"\x00\x00\x00\x01" in data[len(data) // 2:] else len(data)
                data = data[:cut] + random_pps(cfg, rng) + data[cut:]
        elif MODE == "multisps":
            # three different sequences under DIFFERENT parameter-set ids, all parameter sets up front (or each in front of
            # its sequence), the slices of sequence k pointed at PPS k: the switch of the active SPS at an IDR picture
            # (storage.c:379-407), and — when the IDR slices of a sequence are dropped — the refusal to switch elsewhere
            heads, bodies = [], []
            for k in range(3):
                ck = h264writer.random_config(seed * 3 + k); ck["n_pics"] = min(ck.get("n_pics", 6), 5)
                wk = h264writer.StreamWriter(**ck)
                dk = wk.build()
                sid, pid = (k, k) if rng.random() < 0.7 else (rng.randrange(32), rng.randrange(256))
                st = [i for i in range(0, len(dk) - 4) if dk[i:i + 4] == b"\x00\x00\x00\x01" and dk[i - 1:i] != b"\x00"] + [len(dk)]
                units = [dk[st[i]:st[i + 1]] for i in range(len(st) - 1)]
                heads.append(h264writer.write_sps(dict(wk.sps, sps_id=sid)) + h264writer.write_pps(dict(wk.pps, pps_id=pid), dict(wk.sps, sps_id=sid)))
                sl = [with_pps_id(u, pid) for u in units[2:]]
                if k and rng.random() < 0.25:
                    while sl and sl[0][4] & 31 == 5:
                        sl.pop(0)                       # the sequence starts without its IDR picture
                bodies.append(b"".join(sl))
            if rng.random() < 0.5:
                data = b"".join(heads) + b"".join(bodies)
            else:
                data = b"".join(h + b for h, b in zip(heads, bodies))
        elif MODE == "multipps":
            # several picture parameter sets under different ids — same syntax-relevant contents, other QP offsets,
            # constrained-intra and deblocking-control settings left alone — and every slice pointed at one of them:
            # between pictures (activation of another PPS with the same SPS, storage.c:379-414) and, in every fifth stream,
            # from slice to slice inside a picture
            import copy
            w = h264writer.StreamWriter(**cfg)
            ids = rng.sample(range(1, 256), rng.randrange(1, 4))
            extra = b""
            for i in ids:
                pp = copy.deepcopy(w.pps); pp["pps_id"] = i
                pp["chroma_qp_offset"] = rng.randrange(-12, 13)
                pp["pic_init_qp"] = 26 + rng.randrange(-3, 4)
                extra += h264writer.write_pps(pp, w.sps)
            starts = [i for i in range(0, len(data) - 4) if data[i:i + 4] == b"\x00\x00\x00\x01" and data[i - 1:i] != b"\x00"] + [len(data)]
            units = [data[starts[i]:starts[i + 1]] for i in range(len(starts) - 1)]
            per_slice = rng.random() < 0.2
            out, cur = [units[0], units[1], extra], 0
            for u in units[2:]:
                t = u[4] & 31
                if t in (1, 5):
                    first_mb, _ = _read_ue(np.unpackbits(np.frombuffer(_rbsp(u[5:9]), dtype=np.uint8)).tolist() + [1] * 8, 0)
                    if per_slice or rng.random() < 0.3:
                        cur = rng.choice([0] + ids)
                    if rng.random() < 0.03:
                        cur = rng.choice([0] + ids + [rng.randrange(256)])      # now and then an id that was never sent
                    u = with_pps_id(u, cur)
                out.append(u)
            data = b"".join(out)
        elif MODE == "bytestream":
            # the Annex B framing itself (reference h264bsdExtractNalUnit, src/h264bsd_byte_stream.c): 3-byte start codes,
            # extra zero bytes before and after NAL units, a stream that begins without a start code, emulation-prevention
            # and forbidden byte patterns (00 00 00, 00 00 02, 00 00 03 xx) at random places inside NAL units
            starts = [i for i in range(0, len(data) - 4) if data[i:i + 4] == b"\x00\x00\x00\x01" and data[i - 1:i] != b"\x00"] + [len(data)]
            parts = [data[starts[i]:starts[i + 1]] for i in range(len(starts) - 1)]
            out = []
            for k, pnal in enumerate(parts):
                body = bytearray(pnal[4:])
                if rng.random() < 0.15 and len(body) > 6:
                    at = rng.randrange(2, len(body) - 3)
                    body[at:at] = rng.choice([b"\x00\x00\x00", b"\x00\x00\x02", b"\x00\x00\x03", b"\x00\x00\x03\x04", b"\x00\x00\x03\x00\x00\x03", b"\x00\x00"])
                sc = rng.choice([b"\x00\x00\x00\x01", b"\x00\x00\x00\x01", b"\x00\x00\x01", b"\x00" * rng.randrange(3, 8) + b"\x01"])
                if k == 0 and rng.random() < 0.1:
                    sc = rng.choice([b"", b"\x00", b"\x00\x00", b"\x01"])
                tail = b"\x00" * rng.randrange(0, 4) if rng.random() < 0.3 else b""
                out.append(sc + bytes(body) + tail)
            data = b"".join(out)
        elif MODE == "nal":
            # 1-6 NAL units of the types a decoder skips (SEI, access unit delimiter, end of sequence / stream, filler,
            # reserved, unspecified, data partitions) between the stream's own — some of them end an access unit
            # (h264bsdCheckAccessUnitBoundary, src/h264bsd_decoder.c / h264bsd_slice_header.c), a few have a forbidden_zero_bit
            starts = [i for i in range(third, len(data) - 4) if data[i:i + 4] == b"\x00\x00\x00\x01" and data[i - 1:i] != b"\x00"] + [len(data)]
            for _ in range(rng.randrange(1, 7)):
                at = rng.choice(starts)
                t = rng.choice([6, 6, 9, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 2, 3, 4, 23, 24, 31, 0])
                hdr = (0x80 if rng.random() < 0.05 else 0) | (rng.randrange(4) << 5) | t
                body = bytes(rng.randrange(1, 256) for _ in range(rng.randrange(0, 30)))     # (no zero bytes: no start codes inside)
                extra = rng.choice([b"\x00\x00\x00\x01", b"\x00\x00\x01"]) + bytes([hdr]) + body
                data = data[:at] + extra + data[at:]
                starts = [x + len(extra) if x >= at else x for x in starts]
        else:                                          # 1-4 slices with random headers between the stream's own NAL units
            starts = [i for i in range(third, len(data) - 4) if data[i:i + 4] == b"\x00\x00\x00\x01" and data[i - 1:i] != b"\x00"] + [len(data)]
            for _ in range(rng.randrange(1, 5)):
                at = rng.choice(starts)
                extra = random_slice(cfg, rng)
                data = data[:at] + extra + data[at:]
                starts = [x + len(extra) if x >= at else x for x in starts]
        r = run_ref(data); o = run_ours(data)
        n_ok += any(t[0] == 2 for t in r[0]); n_pics += len(r[1])
        if r != o:
            bad.append(seed)
            print("MISMATCH", seed, "trace equal", r[0] == o[0], "pictures equal", r[1] == o[1], "info equal", r[2] == o[2], flush=True)
            if r[0] != o[0]:
                for i, (x, y) in enumerate(zip(r[0], o[0])):
                    if x != y:
                        print("   first trace difference at call", i, x, y, flush=True); break
        if (seed - first) % 500 == 499:
            print("...", seed - first + 1, len(bad), flush=True)
        continue
    data = random_sps(cfg, rng) + data[second:]
    if rng.random() < 0.2:                         # ... and the original SPS again later: a change of parameter sets mid-stream
        cut = data.index(b"\x00\x00\x00\x01", len(data) // 2) if b"\x00\x00\x00\x01" in data[len(data) // 2:] else len(data)
        data = data[:cut] + random_sps(cfg, rng) + data[cut:]
    r = run_ref(data); o = run_ours(data)
    n_ok += any(t[0] == 2 for t in r[0]); n_pics += len(r[1])
    if r != o:
        bad.append(seed)
        print("MISMATCH", seed, "trace equal", r[0] == o[0], "pictures equal", r[1] == o[1], "info equal", r[2] == o[2], flush=True)
        if r[2] != o[2]: print("   ref ", r[2], "\n   ours", o[2], flush=True)
        if r[0] != o[0]:
            for i, (x, y) in enumerate(zip(r[0], o[0])):
                if x != y:
                    print("   first trace difference at call", i, x, y, flush=True); break
    if (seed - first) % 500 == 499:
        print("...", seed - first + 1, len(bad), flush=True)
print(f"{MODE.upper()} sweep {first}..{first + count - 1}: {count - len(bad)} identical, {len(bad)} not {bad[:20]}; headers accepted in {n_ok} streams, "
      f"{n_pics} pictures, {time.time() - t0:.0f} s")
