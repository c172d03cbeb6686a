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
