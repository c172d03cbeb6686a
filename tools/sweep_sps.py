#!/usr/bin/env python3
"""Parity sweep over the SEQUENCE PARAMETER SET: the SPS of a writer-made stream is replaced by one written field by
field with random contents — cropping rectangles, the whole VUI (aspect ratio incl. extended SAR, overscan, video signal
type, chroma location, timing, NAL / VCL HRD with several CPBs, bitstream restriction), levels, constraint flags — with
boundary and out-of-range values mixed in (the reference rejects e.g. max_bytes_per_pic_denom > 16, cpb_cnt > 32, a
cropping rectangle larger than the picture).  Compared with the compiled reference: the h264bsdDecode call trace, the
output pictures, and what the information calls return once headers are ready (h264bsdPicWidth / Height,
CroppingParams, VideoRange, MatrixCoefficients, SampleAspectRatio, Profile, CheckValidParamSets).  TEST TOOL (uses
oracle/).   usage: sweep_sps.py <first seed> <count>"""
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


first, count = int(sys.argv[1]), int(sys.argv[2]); bad = []; t0 = time.time(); n_ok = n_pics = 0
for seed in range(first, first + count):
    rng = random.Random(seed)
    cfg = h264writer.random_config(seed)
    cfg["n_pics"] = min(cfg.get("n_pics", 6), 4)
    data = h264writer.StreamWriter(**cfg).build()
    # the writer's stream starts with its SPS: find the second start code and splice
    second = data.index(b"\x00\x00\x00\x01", 4)
    assert data[4] & 31 == 7
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
print(f"SPS sweep {first}..{first + count - 1}: {count - len(bad)} identical, {len(bad)} not {bad[:20]}; headers accepted in {n_ok} streams, "
      f"{n_pics} pictures, {time.time() - t0:.0f} s")
