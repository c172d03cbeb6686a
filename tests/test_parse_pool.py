"""Host parse pipeline: h264bsdmiDecodePicture / h264bsdmiDecodePictureBatch run the
reference harness loop (posix/test_h264bsd.c:146-177) inside the library, for many instances in parallel on its
parser threads.  The frame jobs they produce must be the ones the plain h264bsdDecode loop produces."""
import hashlib

import pytest

from conftest import stream_bytes


def job_hashes_serial(built, data):
    jobs, _, _ = built.capture_stream(data)
    return [hashlib.sha1(j).hexdigest() for j in jobs]


@pytest.mark.parametrize("threads", [1, 4])
def test_batch_parse_equals_serial_parse(built, threads):
    names = ["test_640x360", "test_1920x1080", "test_640x360", "test_1920x1080_fullRange", "test_640x360", "test_640x360"]
    want = {n: job_hashes_serial(built, stream_bytes(n)) for n in set(names)}
    assert built.lib().h264bsdmiSetParserThreads(threads) >= 1
    got = [[] for _ in names]
    decs = [built.Decoder(capture=(lambda b, k=k: got[k].append(hashlib.sha1(b).hexdigest()))) for k in range(len(names))]
    drv = built.BatchDriver(decs, [stream_bytes(n) for n in names])
    rounds = 0
    while drv.step():
        rounds += 1
    assert rounds == 73
    for k, n in enumerate(names):
        assert got[k] == want[n], f"instance {k} ({n}) produced different frame jobs under the parser pool"
    for d in decs:
        d.close()


def test_decode_picture_loop_counts_errors_and_conceals(built):
    """h264bsdmiDecodePicture on a damaged stream: the same pictures (concealment included) and as many error returns as
    the plain h264bsdDecode loop sees (tests/golden/synth_golden.json holds the reference's trace for the stream)"""
    import ctypes
    import json
    import os
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from damage import damage
    from h264writer import StreamWriter
    from synth_configs import DAMAGED
    name = "damaged_7"
    cfg, dmg = DAMAGED[name]
    data = damage(StreamWriter(**cfg).build(), **dmg)
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "synth_golden.json")))[name]
    L = built.lib()
    jobs = []
    dec = built.Decoder(capture=jobs.append)
    buf = ctypes.create_string_buffer(data, len(data))
    off, n_err, n_pic = 0, 0, 0
    consumed, errs = ctypes.c_uint32(), ctypes.c_uint32()
    for _ in range(10000):
        if off >= len(data):
            break
        st = L.h264bsdmiDecodePicture(dec._st, ctypes.addressof(buf) + off, len(data) - off, n_pic, ctypes.byref(consumed),
                                      ctypes.byref(errs))
        off += consumed.value
        n_err += errs.value
        n_pic += st == built.H264BSD_PIC_RDY
    dec.close()
    assert n_err == sum(1 for t in gold["trace"] if t[0] >= 3)
    assert n_pic == sum(1 for t in gold["trace"] if t[0] == 1)


def test_shared_read_only_input_buffer(built):
    """h264bsdDecode() removes emulation-prevention bytes IN the caller's buffer like the reference does, so by
    default every instance needs a private writable copy.  With h264bsdmiSetInputReadOnly() two instances parse ONE
    read-only mapping of the stream (a write would fault) and produce the frame jobs of an ordinary decode."""
    import ctypes
    import mmap
    import os
    from conftest import ROOT
    data = open(os.path.join(ROOT, "tests", "golden", "test_640x360.h264"), "rb").read()
    assert b"\x00\x00\x03" in data                                   # the stream does contain emulation-prevention bytes
    want, want_trace, _ = built.capture_stream(data)
    m = mmap.mmap(-1, len(data))
    m.write(data)
    addr = ctypes.addressof(ctypes.c_char.from_buffer(m))
    libc = ctypes.CDLL(None, use_errno=True)
    page = mmap.PAGESIZE
    assert addr % page == 0
    assert libc.mprotect(ctypes.c_void_p(addr), ctypes.c_size_t((len(data) + page - 1) // page * page), mmap.PROT_READ) == 0
    try:
        got = [[], []]
        decs = [built.Decoder(capture=got[i].append) for i in range(2)]
        offs, traces = [0, 0], [[], []]
        for d in decs:
            assert built.lib().h264bsdmiSetInputReadOnly(d._st, 1) == 0
        while any(o < len(data) for o in offs):                      # interleaved: both instances walk the same bytes
            for i, d in enumerate(decs):
                if offs[i] < len(data):
                    r, rb = d.decode(addr + offs[i], len(data) - offs[i])
                    traces[i].append((r, rb))
                    offs[i] += rb
                    assert r < built.H264BSD_ERROR
        for d in decs:
            d.close()
        assert got[0] == want and got[1] == want
        assert traces[0] == want_trace and traces[1] == want_trace
    finally:
        libc.mprotect(ctypes.c_void_p(addr), ctypes.c_size_t((len(data) + page - 1) // page * page), mmap.PROT_READ | mmap.PROT_WRITE)


@pytest.mark.parametrize("threads", [3, 0])       # 0: the library's default (batches that pull then run on the CPUs of quota, api.c run_batch)
def test_pull_and_decode_batch_parses_like_the_harness_loop(built, threads):
    """h264bsdmiPullAndDecodePictureBatch (capture mode: the pull pops the output queue, there are no pixels): which frame buffer a
    picture is decoded into depends on what has left the output queue (src/h264bsd_dpb.c: a picture waiting for display keeps its
    buffer; the next slice discards the queue, :1260-1261), so the frame jobs only equal those of the reference harness's loop —
    decode, drain the queue, decode — if an instance with pictures still waiting is not fed"""
    import ctypes
    import os
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from h264writer import StreamWriter
    from synth_configs import CONFIGS
    datas = [StreamWriter(**CONFIGS[n]).build() for n in ("poc0_display_reorder", "everything", "mmco_long_term", "multi_ref")] + [stream_bytes("test_640x360")]

    def harness_loop(data):
        jobs = []
        dec = built.Decoder(capture=lambda b: jobs.append(hashlib.sha1(b).hexdigest()))
        buf = ctypes.create_string_buffer(data, len(data))
        off, stall, outs = 0, 0, []
        while off < len(data):
            r, rb = dec.decode(ctypes.addressof(buf) + off, len(data) - off, len(jobs))
            off += rb
            if r == 1:
                while True:
                    o = dec.next_output_info()
                    if o is None:
                        break
                    outs.append(o[1])
            stall = stall + 1 if rb == 0 else 0
            if stall > 3 or r >= 3:
                break
        dec.close()
        return jobs, outs
    want = [harness_loop(d) for d in datas]
    assert built.lib().h264bsdmiSetParserThreads(threads) >= 1
    got = [[] for _ in datas]
    decs = [built.Decoder(capture=(lambda b, k=k: got[k].append(hashlib.sha1(b).hexdigest()))) for k in range(len(datas))]
    drv = built.BatchDriver(decs, datas)
    held_back = 0
    for _ in range(4000):
        before = list(drv.off)
        drv.step(pull=True)
        held_back += sum(1 for k in range(len(datas)) if before[k] < drv.size[k] and drv.off[k] == before[k])
        if all(o >= s for o, s in zip(drv.off, drv.size)):
            break
    else:
        raise AssertionError("the streams were not consumed")
    for k in range(len(datas)):
        assert got[k] == want[k][0], k
    assert held_back > 0
    for d in decs:
        d.close()
