"""GPU: the drop-in h264bsd C API end to end (host parse -> frame jobs -> HIP engine -> host frames),
driven exactly like /root/reference/posix/test_h264bsd.c:146-177."""
import ctypes
import hashlib

import pytest

from conftest import STREAMS, stream_bytes

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", STREAMS)
def test_decode_loop_matches_reference(name, built, golden):
    g = golden[name]
    dec = built.Decoder()
    shas = []
    trace = dec.decode_stream(stream_bytes(name), on_picture=lambda f, pid, idr, err: shas.append(
        (hashlib.sha256(f.tobytes()).hexdigest(), idr, err)))
    assert [list(t) for t in trace] == g["trace"]
    assert [s[0] for s in shas] == g["frame_sha256"]
    assert [s[1] for s in shas] == [1 if i in (0, 40) else 0 for i in range(73)]
    assert all(s[2] == 0 for s in shas)
    dec.close()


def test_bgra_output_matches_reference(built, golden):
    name = "test_640x360"
    g = golden[name]
    dec = built.Decoder()
    data = stream_bytes(name)
    buf = ctypes.create_string_buffer(data, len(data))
    base, off, n = ctypes.addressof(buf), 0, 0
    while off < len(data) and n < 2:
        r, rb = dec.decode(base + off, len(data) - off)
        off += rb
        if r == built.H264BSD_PIC_RDY:
            pic = dec.next_output_picture_converted(1)
            assert hashlib.sha256(pic[0].tobytes()).hexdigest() == g["convert_sha256"][str(n)][1]
            n += 1
    dec.close()


def test_many_instances_round_robin_are_batched_and_exact(built, golden):
    """8 decoder instances advanced in lock step: every flush reconstructs 8 pictures in one tick"""
    name = "test_640x360"
    g = golden[name]
    data = stream_bytes(name)
    N = 8
    decs = [built.Decoder() for _ in range(N)]
    bufs = [ctypes.create_string_buffer(data, len(data)) for _ in range(N)]
    offs = [0] * N
    pics = [0] * N
    while any(o < len(data) for o in offs):
        ready = []
        for k, d in enumerate(decs):
            while offs[k] < len(data):
                r, rb = d.decode(ctypes.addressof(bufs[k]) + offs[k], len(data) - offs[k])
                offs[k] += rb
                assert r < built.H264BSD_ERROR
                if r == built.H264BSD_PIC_RDY:
                    ready.append(k)
                    break
        for k in ready:          # first pull flushes the whole batch
            f = decs[k].next_output_picture()
            assert hashlib.sha256(f[0].tobytes()).hexdigest() == g["frame_sha256"][pics[k]]
            pics[k] += 1
    assert pics == [73] * N
    for d in decs:
        d.close()
