"""CPU gate: host parser (product, capture mode) + CPU oracle reproduce the REFERENCE bit-exactly.

This is what pins the oracle (and the parser) before any kernel is trusted: golden.json was produced by
the compiled reference (tests/golden/make_golden.py)."""
import os
import hashlib

import pytest

from conftest import STREAMS, stream_bytes
from oracle import pyoracle


@pytest.mark.parametrize("name", STREAMS)
def test_call_trace_matches_reference(name, captured, golden):
    jobs, trace, info = captured(name)
    g = golden[name]
    assert [list(t) for t in trace] == g["trace"]          # same return codes and readBytes, call by call
    assert len(jobs) == g["n_pics"] == 73
    assert (info["width_mbs"], info["height_mbs"]) == (g["width_mbs"], g["height_mbs"])


@pytest.mark.parametrize("name", STREAMS)
def test_parser_plus_oracle_is_bit_exact(name, captured, golden):
    jobs, _, _ = captured(name)
    g = golden[name]
    dpb = pyoracle.OracleDpb(jobs[0])
    total = hashlib.sha256()
    for i, job in enumerate(jobs):
        frame = dpb.decode(job)
        assert hashlib.sha256(frame.tobytes()).hexdigest() == g["frame_sha256"][i], f"{name} frame {i}"
        if i in (0, 1, 40, 72):
            assert pyoracle.checksum64(frame) == g["frame_checksum64"][i]
        total.update(frame.tobytes())
    assert total.hexdigest() == g["sha256_all"]


def test_oracle_colour_conversion_matches_reference(captured, golden):
    name = "test_640x360"
    jobs, _, info = captured(name)
    w, h = info["width_mbs"] * 16, info["height_mbs"] * 16
    dpb = pyoracle.OracleDpb(jobs[0])
    for i, job in enumerate(jobs[:2]):
        frame = dpb.decode(job)
        for fmt in range(3):
            got = hashlib.sha256(pyoracle.oracle_convert(fmt, w, h, frame).tobytes()).hexdigest()
            assert got == golden[name]["convert_sha256"][str(i)][fmt]


def test_getters_match_stream_headers(captured):
    _, _, info = captured("test_1920x1080")
    assert info["cropping"] == (1, 0, 1920, 0, 1080)       # 8 luma rows cropped at the bottom
    assert info["profile"] == 66
    _, _, info_fr = captured("test_1920x1080_fullRange")
    assert info_fr["video_range"] == 1


def test_frame_job_statistics(captured):
    jobs, _, _ = captured("test_1920x1080")
    h0 = pyoracle.blob_header(jobs[0])
    assert h0["is_idr"] == 1 and h0["n_inter"] == 0 and h0["n_intra"] == 8160
    assert h0["n_intra_levels"] <= 120 + 2 * 67                # 2:1 wavefront bound
    h1 = pyoracle.blob_header(jobs[1])
    assert h1["n_inter"] + h1["n_intra"] == 8160
    assert sum(pyoracle.blob_header(j)["n_inter"] for j in jobs) == 410704 + 151131


def _sections(job):
    import struct
    import numpy as np
    h = pyoracle.blob_header(job)
    copy_off, n_copy, gen_off, n_gen, dbk_off, n_dbk = struct.unpack_from("<IIIIII", job, 60)
    n = h["n_mbs"]
    rec = np.frombuffer(job, dtype=np.uint8, count=n * 32, offset=h["rec_off"]).reshape(n, 32)
    lvl = np.frombuffer(job, dtype="<u4", count=h["n_intra_levels"] + 1, offset=h["lvl_off"])
    idx = np.frombuffer(job, dtype="<u2", count=h["n_intra"], offset=h["idx_off"])
    copy = np.frombuffer(job, dtype=np.uint8, count=n_copy * 8, offset=copy_off).reshape(n_copy, 8)
    gen = np.frombuffer(job, dtype="<u2", count=n_gen * 8, offset=gen_off).reshape(n_gen, 8)[:, 0]   # FjGen.mb
    dbk = np.frombuffer(job, dtype="<u2", count=n_dbk, offset=dbk_off)
    return h, rec, lvl, idx, copy, gen, dbk


@pytest.mark.parametrize("pic", [0, 1, 20, 40, 72])
def test_frame_job_schedules_are_consistent(pic, captured):
    """host logic: the work lists the kernels are driven by partition the macroblocks correctly"""
    import numpy as np
    jobs, _, _ = captured("test_1920x1080")
    h, rec, lvl, idx, copy, gen, dbk = _sections(jobs[pic])
    n, w = h["n_mbs"], h["width_mbs"]
    kind = rec[:, 0]
    intra = np.nonzero((kind >= 1) & (kind <= 3))[0]
    inter = np.nonzero(kind == 0)[0]
    # intra schedule: every intra MB exactly once, level starts monotone, neighbours strictly earlier
    assert lvl[0] == 0 and lvl[-1] == len(idx) == len(intra) and (np.diff(lvl.astype(np.int64)) >= 0).all()
    assert sorted(idx.tolist()) == intra.tolist()
    level_of = np.full(n, -1)
    for l in range(len(lvl) - 1):
        level_of[idx[lvl[l]:lvl[l + 1]]] = l
    def needed(r):          # neighbours whose samples the MB's prediction modes read (A=1 B=2 C=4 D=8)
        kind, avail, pred = int(r[0]), int(r[3]), int(r[4])
        need = 0
        if kind == 2:
            need |= (2, 1, 3, 11)[pred & 3]
        elif kind == 1:
            for z in range(16):
                bx, by = ((z >> 2) & 1) * 2 + (z & 1), (z >> 3) * 2 + ((z >> 1) & 1)
                if bx and by:
                    continue
                m = (int(r[24 + (z >> 1)]) >> ((z & 1) * 4)) & 15
                left, top, corner = m in (1, 2, 4, 5, 6, 8), m in (0, 2, 3, 4, 5, 6, 7), m in (4, 5, 6)
                if bx == 0 and (left or corner):
                    need |= 1
                if by == 0 and (top or corner):
                    need |= 2
                if bx == 0 and by == 0 and corner:
                    need |= 8
                if bx == 3 and by == 0 and m in (3, 7):
                    need |= 4
        if kind != 3:
            need |= (3, 1, 2, 11)[(pred >> 2) & 3]
        return need & avail

    for a in intra[:: max(1, len(intra) // 500)]:
        x, y = a % w, a // w
        need = needed(rec[a])
        for bit, ok, nb in ((1, x > 0, a - 1), (2, y > 0, a - w), (4, y > 0 and x + 1 < w, a - w + 1), (8, y > 0 and x > 0, a - w - 1)):
            if ok and (need & bit) and level_of[nb] >= 0:
                assert level_of[nb] < level_of[a]
    # inter MBs: copy runs + general index partition them
    run_mb = copy[:, 0].astype(int) | (copy[:, 1].astype(int) << 8)
    run_cnt = copy[:, 3].astype(int)
    covered = np.concatenate([np.arange(m, m + c) for m, c in zip(run_mb, run_cnt)] + [gen.astype(int)]) if len(inter) else np.array([], int)
    assert sorted(covered.tolist()) == inter.tolist()
    # runs are at most FJ_COPY_RUN (8) macroblocks; only a run with zero displacement may continue into the next row
    # (tiles are contiguous in address order), a displaced one stays inside its row
    displaced = (copy[:, 4:8].view(np.int16).reshape(-1, 2) != 0).any(1) if len(copy) else np.zeros(0, bool)
    assert ((run_cnt >= 1) & (run_cnt <= 8)).all() and (((run_mb % w) + run_cnt <= w) | ~displaced).all()
    # deblocking index: exactly the MBs not marked trivially strength-free
    assert sorted(dbk.tolist()) == np.nonzero(rec[:, 21] == 0)[0].tolist()


def test_input_buffer_is_unescaped_in_place_like_the_reference(built):
    """The reference removes the emulation-prevention bytes of every NAL unit it extracts IN the caller's buffer
    (src/h264bsd_byte_stream.c:193-235, README.md:11).  A drop-in has to leave the buffer in the same state."""
    import ctypes
    import hashlib
    from oracle import pyoracle
    from h264bsd_amd import capi
    from h264writer import StreamWriter
    from synth_configs import CONFIGS
    if not os.path.exists(pyoracle.REF_SO):
        pytest.skip("oracle/_ref not built")
    streams = [stream_bytes("test_640x360"), StreamWriter(**CONFIGS["ipcm"]).build(), StreamWriter(**CONFIGS["high_qp"]).build()]
    n_changed = 0
    for data in streams:
        lib = pyoracle.RefDecoder().lib
        rbuf = ctypes.create_string_buffer(data, len(data))
        dec = lib.h264bsdAlloc()
        assert lib.h264bsdInit(dec, 0) == 0
        off, rb = 0, ctypes.c_uint32(0)
        a, b, c = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32()
        while off < len(data):
            r = lib.h264bsdDecode(dec, ctypes.addressof(rbuf) + off, len(data) - off, 0, ctypes.byref(rb))
            off += rb.value
            if r == 1:
                while lib.h264bsdNextOutputPicture(dec, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)):
                    pass
        lib.h264bsdShutdown(dec)
        lib.h264bsdFree(dec)
        obuf = ctypes.create_string_buffer(data, len(data))
        ours = capi.Decoder(0, capture=lambda blob: None)
        off = 0
        while off < len(data):
            r, n = ours.decode(ctypes.addressof(obuf) + off, len(data) - off, 0)
            off += n
            if r == 1:
                while ours.next_output_info() is not None:
                    pass
        ours.close()
        assert hashlib.sha1(obuf.raw).hexdigest() == hashlib.sha1(rbuf.raw).hexdigest()
        n_changed += obuf.raw != data
    assert n_changed >= 1, "none of the streams contains an emulation-prevention byte: the test checks nothing"
