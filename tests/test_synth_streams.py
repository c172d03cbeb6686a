"""Corner features of the baseline profile on synthetic streams.

tests/h264writer.py writes random but valid streams that use what the bundled x264 streams never do: several
slices per picture, arbitrary slice order, every slice-group map type, I_PCM, constrained intra prediction,
disable_deblocking_filter_idc 0/1/2 with offsets, several reference frames with list reordering, adaptive marking
(MMCO 1-6, long-term frames), frame_num gaps, POC types 0/1/2 with display reordering, non-reference pictures,
repeated IDRs.  The expected answers (h264bsdDecode call trace, output order, picId/isIdr/numErrMbs and a hash of
every output frame) come from the compiled reference decoder (tests/golden/make_synth_golden.py).

CPU test: host parser + CPU oracle.  GPU test: the product (h264bsdInit + HIP engine) through the C ABI."""
import ctypes
import hashlib
import json
import os

import numpy as np
import pytest

import synth
from h264bsd_amd import capi
from h264writer import StreamWriter
from synth_configs import CONFIGS

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "synth_golden.json")))
_streams = {}


def stream_of(name):
    if name not in _streams:
        data = StreamWriter(**CONFIGS[name]).build()
        if hashlib.sha1(data).hexdigest() != GOLD[name]["stream_sha1"]:
            # a FAILURE, not a skip: a run in which 65 streams silently drop out must not read "green" 
            pytest.fail("the writer produced a different stream than the one the golden answers were made from "
                        "(numpy random stream changed?) — regenerate with tests/golden/make_synth_golden.py")
        _streams[name] = data
    return _streams[name]


def check(name, backend):
    data = stream_of(name)
    trace, pics = synth.decode_ours(data, backend)
    g = GOLD[name]
    assert [list(t) for t in trace] == g["trace"], "h264bsdDecode call trace differs from the reference"
    assert [p[1:] for p in pics] == [tuple(p[1:]) for p in g["pics"]], "output order / picId / isIdr / numErrMbs differ"
    bad = [i for i, (p, q) in enumerate(zip(pics, g["pics"])) if p[0] != q[0]]
    assert not bad, f"output pictures {bad} are not bit-exact"


@pytest.mark.parametrize("name", list(CONFIGS))
def test_parser_and_oracle_match_reference(built, name):
    check(name, "oracle")


def test_golden_matches_live_reference(built):
    """the committed answers are what oracle/_ref says today (a few streams; skipped where _ref is absent)"""
    from oracle import pyoracle
    if not os.path.exists(pyoracle.REF_SO):
        pytest.skip("oracle/_ref not built")
    for name in ("everything", "gaps_full_dpb", "mmco_long_term", "fmo_explicit"):
        trace, pics = synth.decode_reference(stream_of(name))
        assert [list(t) for t in trace] == GOLD[name]["trace"]
        assert [list(p) for p in pics] == GOLD[name]["pics"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CONFIGS))
def test_gpu_matches_reference(built, name):
    check(name, "gpu")


@pytest.mark.parametrize("name", ["poc0_display_reorder", "mmco_long_term", "everything", "frame_num_gaps", "poc1_nonref_idr"])
def test_no_output_reordering_mode_matches_live_reference(built, name):
    """h264bsdInit(storage, noOutputReordering = 1): DPB of max(num_ref_frames, 1) frames, pictures are output in
    decoding order (reference src/h264bsd_dpb.c:1014-1040, :806-815); compared with oracle/_ref directly"""
    from oracle import pyoracle
    if not os.path.exists(pyoracle.REF_SO):
        pytest.skip("oracle/_ref not built")
    data = stream_of(name)
    assert synth.decode_ours(data, "oracle", 1) == synth.decode_reference(data, 1)


SEQUENCE_CHANGES = [("multi_ref", "fmo_dispersed"), ("fmo_dispersed", "multi_ref", "poc0_display_reorder"),
                    ("poc0_display_reorder", "fmo_dispersed", "poc0_display_reorder"), ("multi_ref", "multi_ref")]


def _concat(names):
    return b"".join(stream_of(n) for n in names)


@pytest.mark.parametrize("names", SEQUENCE_CHANGES, ids=["+".join(n) for n in SEQUENCE_CHANGES])
def test_sequence_parameter_set_changes(built, names):
    """a new SPS at an IDR (other picture size / DPB size): second HDRS_RDY, DPB re-initialised, pictures still waiting
    for output are lost exactly as in the reference (src/h264bsd_storage.c:297-419, h264bsdResetDpb)"""
    from oracle import pyoracle
    if not os.path.exists(pyoracle.REF_SO):
        pytest.skip("oracle/_ref not built")
    data = _concat(names)
    assert synth.decode_ours(data, "oracle") == synth.decode_reference(data)


@pytest.mark.gpu
@pytest.mark.parametrize("names", SEQUENCE_CHANGES, ids=["+".join(n) for n in SEQUENCE_CHANGES])
def test_gpu_sequence_parameter_set_changes(built, names):
    """the engine re-allocates the stream's frame buffers at the second activation"""
    data = _concat(names)
    assert synth.decode_ours(data, "gpu") == synth.decode_ours(data, "oracle")


def _live_sweep(first, count, damaged, backend, concat=1):
    """fresh random streams against the LIVE reference (oracle/_ref travels to the GPU box as a built artefact);
    concat > 1: several different streams one after the other — parameter sets with the same ids and other picture / DPB
    sizes, the next sequence sometimes starting in the middle of the previous one"""
    from oracle import pyoracle
    import damage as dmg
    from h264writer import random_config
    if not os.path.exists(pyoracle.REF_SO):
        pytest.skip("oracle/_ref not built")
    bad = []
    for seed in range(first, first + count):
        parts = []
        for k in range(concat):
            sub = seed if concat == 1 else seed * concat + k
            part = StreamWriter(**random_config(sub)).build()
            if damaged:
                part = dmg.damage(part, sub, p_drop=0.15, p_flip=0.25, p_trunc=0.15)
            if k + 1 < concat and (seed + k) & 1:
                part = part[: len(part) * 2 // 3]
            parts.append(part)
        data = b"".join(parts)
        nor = 0 if damaged else seed & 1
        if synth.decode_reference(data, nor) != synth.decode_ours(data, backend, nor):
            bad.append(seed)
    assert not bad, f"seeds {bad} differ from the reference"


def test_random_streams_match_live_reference(built):
    _live_sweep(91000, 40, False, "oracle")
    _live_sweep(91500, 60, True, "oracle")
    _live_sweep(93000, 30, True, "oracle", concat=3)


@pytest.mark.gpu
def test_gpu_random_streams_match_live_reference(built):
    _live_sweep(92000, 120, False, "gpu")
    _live_sweep(92500, 160, True, "gpu")
    _live_sweep(93500, 60, True, "gpu", concat=3)     # the engine re-allocates (zeroed) frame buffers at every activation


@pytest.mark.gpu
def test_gpu_pull_and_decode_batch_follows_the_call_protocol(built):
    """h264bsdmiPullAndDecodePictureBatch on streams whose pictures leave in another order than they are decoded (display
    reordering, MMCO, frame_num gaps, a new SPS): every round pulls ONE picture per instance and then parses on — unless the
    instance still has pictures waiting, which the next slice would discard (src/h264bsd_dpb.c:1260-1261): then it is not fed.
    Same pictures, ids, IDR flags and error counts as the one-call-at-a-time loop of the reference harness."""
    names = ["poc0_display_reorder", "everything", "multi_ref", "mmco_long_term", "frame_num_gaps", "poc1_nonref_idr", "fmo_dispersed"]
    datas = [stream_of(n) for n in names] + [_concat(("fmo_dispersed", "multi_ref", "poc0_display_reorder"))]
    want = [synth.decode_ours(d, "gpu")[1] for d in datas]
    capi.lib().h264bsdmiSetParserThreads(4)
    decs = [capi.Decoder() for _ in datas]
    drv = capi.BatchDriver(decs, datas)
    got = [[] for _ in datas]

    def take(k, n_bytes, ptr, pid, idr, nerr):
        frame = np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_uint8)), (n_bytes,))
        got[k].append((hashlib.sha1(frame.tobytes()).hexdigest(), pid, idr, nerr))
    held_back = 0
    for _ in range(4000):
        before = list(drv.off)
        size_before = [d.frame_bytes() for d in decs]       # a pulled picture has the size its instance reported BEFORE the call: the parsing
        drv.step(pull=True)                                  # behind the pull may activate another sequence parameter set
        for k, p in drv.pulled.items():
            take(k, size_before[k], *p)
        held_back += sum(1 for k in range(len(datas)) if before[k] < drv.size[k] and drv.off[k] == before[k] and k in drv.pulled)
        if all(o >= s for o, s in zip(drv.off, drv.size)):
            break
    else:
        raise AssertionError("the streams were not consumed")
    for k, d in enumerate(decs):
        d.flush_buffer()
        while True:
            o = d.next_output_picture()
            if o is None:
                break
            got[k].append((hashlib.sha1(np.ascontiguousarray(o[0]).tobytes()).hexdigest(), o[1], o[2], o[3]))
        d.close()
    for k in range(len(datas)):
        assert got[k] == [tuple(p) for p in want[k]], k
    assert held_back > 0            # the reordering streams did hand out several pictures after one decode
