"""Damaged streams: lost and truncated slices.

The reference reports H264BSD_ERROR for the broken NAL unit, un-decodes the macroblocks of the corrupt slice
(src/h264bsd_slice_data.c:298-354), conceals every macroblock that is still missing when the access unit ends
(src/h264bsd_conceal.c) — copy from the first usable reference in P pictures, synthesis from the neighbours otherwise,
filtered as intra with QP 40 — delivers the picture with numErrMbs set and keeps decoding.  These tests require the
same call trace, the same output order, the same numErrMbs and bit-identical pictures (concealed pixels included)
for 48 damaged synthetic streams; expected answers from the compiled reference (tests/golden/make_synth_golden.py).

CPU test: host parser (planning) + CPU oracle.  GPU test: the product through the C ABI (concealment runs in the
HIP kernels: copies in k_copy, synthesis in k_frame_intra)."""
import hashlib
import json
import os

import pytest

import synth
from damage import damage
from h264writer import StreamWriter
from synth_configs import DAMAGED, DAMAGED_BUNDLED, FLIPPED, OVERFLOW, REDUNDANT, SWEEP_FINDS

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "synth_golden.json")))
# streams on which the reference shows or predicts from frame-buffer memory it never wrote (found by
# tests/golden/make_synth_golden.py with two heap fill bytes): the golden answers are those of the reference with its
# allocations starting out zeroed (synth.decode_reference), as this repository's frame buffers do
HEAP_DEPENDENT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_undefined.json")))
UNDEFINED = ()          # nothing is skipped
ALL = {**DAMAGED, **FLIPPED, **OVERFLOW, **REDUNDANT, **SWEEP_FINDS}
NAMES = list(ALL) + list(DAMAGED_BUNDLED)
_streams = {}


def stream_of(name):
    if name not in _streams:
        if name in DAMAGED_BUNDLED:
            stream, dmg = DAMAGED_BUNDLED[name]
            data = damage(open(os.path.join(os.path.dirname(__file__), "golden", stream + ".h264"), "rb").read(), **dmg)
        else:
            cfg, dmg = ALL[name]
            data = damage(StreamWriter(**cfg).build(), **dmg)
        if hashlib.sha1(data).hexdigest() != GOLD[name]["stream_sha1"]:
            # a FAILURE, not a skip: ~230 damaged streams silently dropping out must not read "green" 
            pytest.fail("stream differs from the one the golden answers were made from — regenerate the golden file")
        _streams[name] = data
    return _streams[name]


def check(name, backend):
    trace, pics = synth.decode_ours(stream_of(name), backend)
    g = GOLD[name]
    assert [list(t) for t in trace] == g["trace"], "h264bsdDecode call trace differs from the reference"
    assert [p[1:] for p in pics] == [tuple(p[1:]) for p in g["pics"]], "output order / picId / isIdr / numErrMbs differ"
    bad = [i for i, (p, q) in enumerate(zip(pics, g["pics"])) if p[0] != q[0]]
    assert not bad, f"output pictures {bad} are not bit-exact"


def test_the_set_really_exercises_concealment():
    n_err = sum(p[3] for name in DAMAGED for p in GOLD[name]["pics"])
    n_calls = sum(1 for name in DAMAGED for t in GOLD[name]["trace"] if t[0] == 3)
    assert n_err > 1500 and n_calls > 100


def test_the_bit_error_and_residual_range_sets_are_not_empty_shells():
    flipped = [n for n in FLIPPED if n not in UNDEFINED]
    overflow = [n for n in OVERFLOW if n not in UNDEFINED]
    assert len(flipped) >= 48 and len(overflow) >= 24
    assert sum(1 for n in flipped for t in GOLD[n]["trace"] if t[0] == 3) > 300          # H264BSD_ERROR calls
    # every residual-range stream is intact as a bitstream: each error call in it is the reference's
    # h264bsdProcessBlock range check (transform.c:184-188), each concealed macroblock its consequence
    assert sum(1 for n in overflow for t in GOLD[n]["trace"] if t[0] == 3) > 100
    assert sum(p[3] for n in overflow for p in GOLD[n]["pics"]) > 300


def test_the_redundant_slice_set_reaches_the_second_decodes(built):
    """The set must really contain macroblocks that are decoded twice: pictures that the parser splits into a
    reconstruction-only job and a deblock-only job (FjHeader.ghost / dbk_only)."""
    import ctypes
    import h264bsd_amd
    from h264bsd_amd import capi
    split = 0
    for name in [n for n in REDUNDANT if n not in UNDEFINED][:16]:
        jobs = []
        dec = capi.Decoder(0, capture=lambda b: jobs.append(h264bsd_amd.job_header(bytes(b))))
        data = stream_of(name)
        buf = ctypes.create_string_buffer(data, len(data))
        base, off, stall, pid = ctypes.addressof(buf), 0, 0, 0
        while off < len(data) and stall <= 3:
            r, rb = dec.decode(base + off, len(data) - off, pid)
            off += rb
            pid += r == 1
            stall = stall + 1 if rb == 0 else 0
        dec.close()
        split += sum(1 for j in jobs if j["dbk_only"])
    assert split >= 20


@pytest.mark.parametrize("name", NAMES)
def test_parser_and_oracle_match_reference(built, name):
    check(name, "oracle")


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_gpu_matches_reference(built, name):
    check(name, "gpu")
