"""GPU parity: the HIP kernels, called through the C ABI, against the reference's golden frames and
against the CPU oracle stage by stage.  Run on the MI355X box:  pytest -m gpu"""
import hashlib

import numpy as np
import pytest

from conftest import STREAMS, stream_bytes
from oracle import pyoracle

pytestmark = pytest.mark.gpu


def _first_diff(a, b, wmb, hmb):
    d = np.nonzero(a != b)[0]
    if d.size == 0:
        return "identical"
    i = int(d[0])
    W, H = wmb * 16, hmb * 16
    if i < W * H:
        return f"{d.size} bytes differ; first: luma x={i % W} y={i // W} (mb {i % W // 16},{i // W // 16}) got {a[i]} want {b[i]}"
    j = i - W * H
    pl = "cb" if j < W * H // 4 else "cr"
    j %= W * H // 4
    return f"{d.size} bytes differ; first: {pl} x={j % (W // 2)} y={j // (W // 2)} (mb {j % (W // 2) // 8},{j // (W // 2) // 8}) got {a[i]} want {b[i]}"


@pytest.mark.parametrize("name", STREAMS)
def test_replay_every_frame_matches_reference(name, built, captured, golden):
    """kernels only (jobs resident in HBM), 2 private stream copies, picture by picture"""
    jobs, _, info = captured(name)
    g = golden[name]
    rep = built.Replay(jobs, n_streams=2)
    try:
        for i, job in enumerate(jobs):
            rep.run(i, 1)
            cur = pyoracle.blob_header(job)["cur_slot"]
            for s in range(2):
                got = rep.fetch(s, cur)
                if hashlib.sha256(got.tobytes()).hexdigest() != g["frame_sha256"][i]:
                    # diagnose against the oracle rendering of the same jobs
                    dpb = pyoracle.OracleDpb(jobs[0])
                    want = None
                    for j in jobs[: i + 1]:
                        want = dpb.decode(j)
                    pytest.fail(f"{name} picture {i} stream {s}: " + _first_diff(got, want, info["width_mbs"], info["height_mbs"]))
            sums = rep.checksums(cur)
            assert int(sums[0]) == int(sums[1]) == g["frame_checksum64"][i]
    finally:
        rep.close()


def test_reconstruction_without_deblocking_matches_oracle(built, captured):
    """stage isolation: inter + intra reconstruction only, against oracle_recon"""
    name = "test_640x360"
    jobs, _, info = captured(name)
    rep = built.Replay(jobs[:12], n_streams=1)
    try:
        rep.set_stages(3)
        dpb = pyoracle.OracleDpb(jobs[0])
        for i, job in enumerate(jobs[:12]):
            rep.run(i, 1)
            want = dpb.decode(job, deblock=False)
            got = rep.fetch(0, pyoracle.blob_header(job)["cur_slot"])
            assert np.array_equal(got, want), f"picture {i}: " + _first_diff(got, want, info["width_mbs"], info["height_mbs"])
    finally:
        rep.close()


def test_whole_stream_in_one_submission(built, captured, golden):
    """all 73 ticks enqueued back to back (what bench.py times), verified at the end"""
    name = "test_1920x1080"
    jobs, _, _ = captured(name)
    g = golden[name]
    rep = built.Replay(jobs, n_streams=3)
    try:
        rep.run()
        rep.sync()
        last = pyoracle.blob_header(jobs[-1])["cur_slot"]
        sums = rep.checksums(last)
        assert [int(x) for x in sums] == [g["frame_checksum64"][-1]] * 3
        t = rep.timings()
        assert t["k_recon_inter"][1] == 71 and t["k_frame_dbk"][1] == 73 and t["total_ms"] > 0
    finally:
        rep.close()


def test_staggered_streams_mix_pictures_in_one_tick(built, captured, golden):
    """odd streams start at the second IDR: every tick mixes two different pictures (an all-intra one with a P
    picture at the IDR ticks), twice around so that the wrap-around onto the first IDR is covered"""
    name = "test_640x360"
    jobs, _, _ = captured(name)
    g = golden[name]
    heads = [pyoracle.blob_header(j) for j in jobs]
    off = [i for i, h in enumerate(heads) if h["is_idr"] and i > 0][0]
    rep = built.Replay(jobs, n_streams=5, odd_offset=off)
    try:
        for lap in range(2):
            for i in range(len(jobs)):
                rep.run(i, 1)
                p = (i + off) % len(jobs)
                even = rep.checksums(heads[i]["cur_slot"])
                odd = rep.checksums(heads[p]["cur_slot"])
                assert [int(x) for x in even[0::2]] == [g["frame_checksum64"][i]] * 3, (lap, i)
                assert [int(x) for x in odd[1::2]] == [g["frame_checksum64"][p]] * 2, (lap, i, p)
    finally:
        rep.close()


@pytest.mark.parametrize("lanes,delay,groups,S", [(0, 0, 1, 11), (2, 2, 1, 11), (3, 5, 1, 11), (2, 3, 4, 37), (3, 4, 9, 40)])
def test_desynchronised_streams(lanes, delay, groups, S, built, captured, golden):
    """every stream at its own picture index (so every tick mixes I pictures and P pictures), with and without
    heavy lanes and stream groups (the lane scheduler of h264bsdmiReplayCreateSched: one heavy launch per round, the
    streams regrouped by estimated cost every 32 rounds, ordering across lanes by events); after each lap stream s has
    just finished picture offsets[s] - 1"""
    name = "test_640x360"
    jobs, _, _ = captured(name)
    g = golden[name]["frame_checksum64"]
    heads = [pyoracle.blob_header(j) for j in jobs]
    n = len(jobs)
    offsets = [(s * n) // S for s in range(S)]
    rep = built.Replay(jobs, n_streams=S, offsets=offsets, heavy_lanes=lanes, heavy_delay=delay, groups=groups)
    try:
        for lap in range(3):
            rep.run()
            rep.sync()
            sums = {slot: rep.checksums(slot) for slot in set(h["cur_slot"] for h in heads)}
            for s in range(S):
                last = (offsets[s] - 1) % n
                assert int(sums[heads[last]["cur_slot"]][s]) == g[last], (lap, s, last)
    finally:
        rep.close()


def test_one_replay_set_under_one_schedule_after_the_other(built, captured, golden):
    """h264bsdmiReplayReschedule: the resident jobs of ONE set replayed lock-step, staggered, desynchronised with heavy lanes
    and stream groups, with 4 lock-step groups, and lock-step again — every schedule bit-exact, like a set created for it
    (bench.py shares one set between its legs this way)"""
    name = "test_640x360"
    jobs, _, _ = captured(name)
    g = golden[name]["frame_checksum64"]
    heads = [pyoracle.blob_header(j) for j in jobs]
    n, S = len(jobs), 12
    slots = set(h["cur_slot"] for h in heads)
    rep = built.Replay(jobs, n_streams=S)

    def lock_step_pass():
        for i in range(n):
            rep.run(i, 1)
            assert [int(x) for x in rep.checksums(heads[i]["cur_slot"])] == [g[i]] * S, i

    def laps(offsets, count):
        for lap in range(count):
            rep.run(); rep.sync()
            sums = {slot: rep.checksums(slot) for slot in slots}
            for s in range(S):
                last = (offsets[s] - 1) % n
                assert int(sums[heads[last]["cur_slot"]][s]) == g[last], (lap, s, last)

    try:
        lock_step_pass()
        off = [i for i, h in enumerate(heads) if h["is_idr"] and i > 0][0]
        rep.reschedule(odd_offset=off)
        for i in range(n):
            rep.run(i, 1)
            p = (i + off) % n
            assert [int(x) for x in rep.checksums(heads[i]["cur_slot"])[0::2]] == [g[i]] * (S // 2), i
            assert [int(x) for x in rep.checksums(heads[p]["cur_slot"])[1::2]] == [g[p]] * (S // 2), (i, p)
        offsets = [(s * n) // S for s in range(S)]
        rep.reschedule(offsets=offsets, heavy_lanes=2, heavy_delay=3, groups=3)
        laps(offsets, 2)
        rep.reschedule(offsets=offsets)                  # common ticks
        laps(offsets, 2)
        rep.reschedule()
        rep.set_groups(4)
        rep.run(); rep.sync()
        assert [int(x) for x in rep.checksums(heads[-1]["cur_slot"])] == [g[-1]] * S
        rep.reschedule()
        lock_step_pass()
    finally:
        rep.close()


@pytest.mark.parametrize("name", ["test_640x360", "test_1920x1080_fullRange"])
def test_on_device_colour_conversion(name, built, captured, golden):
    jobs, _, info = captured(name)
    g = golden[name]
    w, h = info["width_mbs"] * 16, info["height_mbs"] * 16
    rep = built.Replay(jobs[:2], n_streams=2)
    try:
        for i in range(2):
            rep.run(i, 1)
            cur = pyoracle.blob_header(jobs[i])["cur_slot"]
            for fmt in range(3):
                rep.convert(cur, fmt)
                got = rep.fetch_converted(1, w * h)
                assert hashlib.sha256(got.tobytes()).hexdigest() == g["convert_sha256"][str(i)][fmt]
    finally:
        rep.close()


def test_stateless_convert_api_random_frames(built):
    rng = np.random.default_rng(7)
    for (w, h) in ((16, 16), (64, 48), (640, 368)):
        yuv = rng.integers(0, 256, w * h * 3 // 2, dtype=np.uint8)
        for fmt in range(3):
            assert np.array_equal(built.convert(fmt, w, h, yuv), pyoracle.oracle_convert(fmt, w, h, yuv))


def test_config4_at_full_size_256_streams(built, captured, golden):
    """BASELINE.json config 4 exactly as the bench runs it — 256 concurrent copies of test_1920x1080.h264, all 73
    pictures, private frame jobs and DPBs per stream (22 GB of HBM) — outside bench.py: after every tick the
    device-computed checksum of the picture it produced is compared, for all 256 streams, with the reference's."""
    name = "test_1920x1080"
    jobs, _, _ = captured(name)
    errs_before = built.device_error_events()      # (a counter, not the sticky bits that the hand-built jobs of test_gpu_random_jobs.py set on purpose)
    rep = built.Replay(jobs, n_streams=256)
    g = golden[name]
    for i, job in enumerate(jobs):
        rep.run(i, 1)
        sums = rep.checksums(pyoracle.blob_header(job)["cur_slot"])
        assert sums.shape == (256,) and (sums == np.uint64(g["frame_checksum64"][i])).all(), f"picture {i}"
    assert built.device_error_events() == errs_before           # no event: frame jobs of the parser never trip a wire
    rep.close()


@pytest.mark.parametrize("name,streams,laps", [("test_1920x1080", 256, 3), ("test_640x360", 48, 6)])
def test_whole_laps_without_a_host_synchronisation_between_ticks(name, streams, laps, built, captured, golden):
    """The lock-step schedule as bench.py times it: all ticks of a lap enqueued back to back — k_copy and k_dbk of tick i + 1 on HIP
    streams of their own right behind tick i's k_frame_dbk (engine.hip launch_tick) — and only then the device-computed checksums
    of the picture every frame-buffer slot was written with last, for all streams.  (The per-tick tests above put a host
    synchronisation between two ticks.)"""
    jobs, _, _ = captured(name)
    g = golden[name]["frame_checksum64"]
    last_in_slot = {}
    for i, job in enumerate(jobs):
        last_in_slot[pyoracle.blob_header(job)["cur_slot"]] = i
    errs_before = built.device_error_events()
    rep = built.Replay(jobs, n_streams=streams)
    try:
        for lap in range(laps):
            rep.run()
            rep.sync()
            for slot, i in last_in_slot.items():
                sums = rep.checksums(slot)
                assert sums.shape == (streams,) and (sums == np.uint64(g[i])).all(), f"lap {lap}: slot {slot}, picture {i}"
        assert built.device_error_events() == errs_before
    finally:
        rep.close()


@pytest.mark.parametrize("name,streams", [("test_640x360", 5), ("test_1920x1080", 3)])
def test_hosted_colour_conversion_equals_the_launch(name, streams, built, captured, golden):
    """Config 3 inside a run: the pictures of tick i - 1 are converted by wavefronts of tick i's k_frame_dbk workgroups
    (kernels/convert.hip.h, conv_drain) — what they leave in the conversion buffer must be what the k_convert_tiles launch
    writes for the same picture, for every picture of the stream, every stream and every format, and the decoded pictures
    must not notice.  The launch itself is pinned to the reference's conversion by test_on_device_colour_conversion."""
    jobs, _, info = captured(name)
    g = golden[name]
    w, h = info["width_mbs"] * 16, info["height_mbs"] * 16
    heads = [pyoracle.blob_header(j) for j in jobs]
    n = len(jobs)
    rep = built.Replay(jobs, n_streams=streams)
    try:
        launches_seen = ticks_seen = 0
        for fmt in (1, 0, 2):
            for k in ([0, 1, 2, 17, n - 2] if fmt != 1 else range(n - 1)):
                # ticks 0 .. k + 1, no launch behind the last one: the buffer holds picture k as tick k + 1's hosts converted it
                rep.set_convert(fmt, trailing=False)
                rep.run(0, k + 2); rep.sync()
                _, launches = rep.convert_timings()
                got = [rep.fetch_converted(s, w * h).copy() for s in range(streams)]
                assert [int(x) for x in rep.checksums(heads[k + 1]["cur_slot"])] == [g["frame_checksum64"][k + 1]] * streams
                rep.set_convert(-1)
                # the same picture through the launch
                rep.run(0, k + 1)
                rep.convert(heads[k]["cur_slot"], fmt)
                for s in range(streams):
                    assert np.array_equal(got[s], rep.fetch_converted(s, w * h)), f"picture {k}, stream {s}, format {fmt}"
                if str(k) in g["convert_sha256"]:
                    assert hashlib.sha256(got[streams - 1].tobytes()).hexdigest() == g["convert_sha256"][str(k)][fmt]
                launches_seen += launches; ticks_seen += k + 1
        assert launches_seen * 2 < ticks_seen          # most pictures really were converted by hosts, not by launches
    finally:
        rep.close()


@pytest.mark.parametrize("cfg", [dict(n_pics=9, wmb=5, hmb=4, seed=41, idc=(1,)),                          # no picture is filtered: no k_frame_dbk launch, k_convert_rest converts
                                 dict(n_pics=10, wmb=6, hmb=5, seed=42, slices_per_pic=1, idc=(0, 1)),        # filtered and unfiltered pictures: a workgroup with nothing to filter converts
                                 dict(n_pics=8, wmb=7, hmb=3, seed=43, idc=(0,))])                            # odd width: the last tile pair of a row has no right-hand tile
def test_hosted_colour_conversion_on_synthetic_streams(cfg, built):
    """the hosted conversion where k_frame_dbk has nothing to do for a picture, or is not launched at all, and on a picture an
    odd number of macroblocks wide: every picture of the stream against the k_convert_tiles launch, which is pinned to the
    oracle's conversion (the reference's formula) here"""
    from h264writer import StreamWriter
    data = StreamWriter(**cfg).build()
    jobs, _, info = built.capture_stream(data)
    w, h = info["width_mbs"] * 16, info["height_mbs"] * 16
    heads = [pyoracle.blob_header(j) for j in jobs]
    if cfg["idc"] == (1,):
        assert not any(hd["any_deblock"] for hd in heads)
    streams = 3
    rep = built.Replay(jobs, n_streams=streams)
    try:
        for fmt in range(3):
            for k in range(len(jobs) - 1):
                if heads[k + 1]["cur_slot"] == heads[k]["cur_slot"]:
                    continue
                rep.set_convert(fmt, trailing=False)
                rep.run(0, k + 2); rep.sync()
                got = [rep.fetch_converted(s, w * h).copy() for s in range(streams)]
                rep.set_convert(-1)
                rep.run(0, k + 1)
                yuv = rep.fetch(streams - 1, heads[k]["cur_slot"])
                want = pyoracle.oracle_convert(fmt, w, h, yuv)
                rep.convert(heads[k]["cur_slot"], fmt)
                for s in range(streams):
                    assert np.array_equal(rep.fetch_converted(s, w * h), want), f"launch: picture {k}, stream {s}, format {fmt}"
                    assert np.array_equal(got[s], want), f"hosted: picture {k}, stream {s}, format {fmt}"
    finally:
        rep.close()
