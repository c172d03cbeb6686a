"""CPU gate: host parser (product, capture mode) + CPU oracle reproduce the REFERENCE bit-exactly.

This is what pins the oracle (and the parser) before any kernel is trusted: golden.json was produced by
the compiled reference (tests/golden/make_golden.py)."""
import hashlib

import pytest

from conftest import STREAMS
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
