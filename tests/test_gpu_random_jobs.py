"""GPU parity on hand-built, randomised frame jobs (tests/jobgen.py): kernels vs CPU oracle, bit-exact.
Covers what the three bundled streams do not: motion vectors far outside the picture (clamp-to-edge), every
fractional position with per-4x4 vectors and several reference slots, I_PCM, every intra mode under arbitrary
availability, all QPs / filter offsets / per-MB deblocking flags, odd picture sizes (1 MB wide/high)."""
import numpy as np
import pytest

from oracle import pyoracle
from jobgen import build_job

pytestmark = pytest.mark.gpu


def _run(built, jobs, n_streams=2, stages=7):
    rep = built.Replay(jobs, n_streams=n_streams)
    rep.set_stages(stages)
    dpb = pyoracle.OracleDpb(jobs[0])
    try:
        for i, job in enumerate(jobs):
            rep.run(i, 1)
            want = dpb.decode(job, deblock=bool(stages & 4))
            cur = pyoracle.blob_header(job)["cur_slot"]
            for s in range(n_streams):
                got = rep.fetch(s, cur)
                if not np.array_equal(got, want):
                    d = np.nonzero(got != want)[0]
                    h = pyoracle.blob_header(job)
                    W = h["width_mbs"] * 16
                    i0 = int(d[0])
                    where = f"luma x={i0 % W} y={i0 // W}" if i0 < W * h["height_mbs"] * 16 else f"chroma byte {i0 - W * h['height_mbs'] * 16}"
                    pytest.fail(f"picture {i} stream {s}: {d.size} bytes differ, first at {where}: got {got[i0]} want {want[i0]}")
    finally:
        rep.close()


@pytest.mark.parametrize("seed,wmb,hmb", [(1, 6, 5), (2, 11, 7), (3, 1, 1), (4, 1, 9), (5, 9, 1), (6, 20, 12), (7, 5, 4)])
def test_random_pictures_full_pipeline(built, seed, wmb, hmb):
    rng = np.random.default_rng(seed)
    lib = built.lib()
    jobs = [build_job(lib, rng, wmb, hmb, 0, 4, [])]                       # intra / PCM only
    jobs.append(build_job(lib, rng, wmb, hmb, 1, 4, [0]))
    jobs.append(build_job(lib, rng, wmb, hmb, 2, 4, [0, 1]))
    jobs.append(build_job(lib, rng, wmb, hmb, 3, 4, [0, 1, 2], p_inter=0.9))
    jobs.append(build_job(lib, rng, wmb, hmb, 0, 4, [1, 2, 3], p_inter=0.97, mv_range=64))
    _run(built, jobs)


@pytest.mark.parametrize("seed", [11, 12])
def test_random_pictures_reconstruction_only(built, seed):
    rng = np.random.default_rng(seed)
    lib = built.lib()
    jobs = [build_job(lib, rng, 8, 6, 0, 3, []), build_job(lib, rng, 8, 6, 1, 3, [0]), build_job(lib, rng, 8, 6, 2, 3, [0, 1])]
    _run(built, jobs, n_streams=1, stages=3)


@pytest.mark.parametrize("seed", [31, 32, 33])
def test_residuals_at_the_16_bit_bound(built, seed):
    """inter macroblocks whose level magnitudes put the parser's bound (hd_resid.c) within +-15 % of its limit: on one side
    k_recon_inter runs luma and chroma as one packed 16-bit transform (intermediates use most of the 16 bits), on the other
    (FJ_CODED_WIDE, set by h264bsdmiJobFinalize for hand-built jobs) in 32 bits; both must equal the oracle's 32-bit arithmetic"""
    import struct
    rng = np.random.default_rng(seed)
    lib = built.lib()
    jobs = [build_job(lib, rng, 9, 6, 0, 3, [])]
    jobs.append(build_job(lib, rng, 9, 6, 1, 3, [0], p_inter=0.95, mv_range=80, near_bound=True))
    jobs.append(build_job(lib, rng, 9, 6, 2, 3, [0, 1], p_inter=0.95, mv_range=80, near_bound=True))
    wide = narrow = 0
    for j in jobs[1:]:
        h = pyoracle.blob_header(j)
        for a in range(h["n_mbs"]):
            kind = j[h["rec_off"] + 32 * a]
            coded = struct.unpack_from("<I", j, h["rec_off"] + 32 * a + 8)[0]
            if kind == 0 and coded & 0x02FFFFFF:
                wide += (coded >> 27) & 1
                narrow += 1 - ((coded >> 27) & 1)
    assert wide >= 10 and narrow >= 10, (wide, narrow)
    _run(built, jobs, stages=3)


def test_no_deblocking_at_all(built):
    rng = np.random.default_rng(21)
    lib = built.lib()
    jobs = [build_job(lib, rng, 7, 5, 0, 2, [], any_deblock=False), build_job(lib, rng, 7, 5, 1, 2, [0], any_deblock=False)]
    _run(built, jobs)


# ---- row bands of the two per-picture kernels (k_frame_dbk / k_frame_intra, kernels.hip.h): a picture split over several
# workgroups with the hand-over through HBM must give the same samples as one workgroup ----
DEFAULT_TAIL = (17, 9, 8, 0, 9, 12, 320)       # engine.hip TailConfig


@pytest.fixture
def tail(built):
    def set_(*cfg):
        built.set_tail(*cfg)
    yield set_
    built.set_tail(*DEFAULT_TAIL)


@pytest.mark.parametrize("rows,waves", [(1, 1), (1, 4), (2, 2), (3, 12)])
@pytest.mark.parametrize("seed,wmb,hmb", [(31, 6, 5), (32, 11, 7), (33, 1, 1), (34, 1, 9), (35, 9, 1), (36, 20, 12)])
def test_random_pictures_in_row_bands(built, tail, seed, wmb, hmb, rows, waves):
    """bands of 1-3 macroblock rows (>= 4 bands wherever the picture has the rows), 1-12 wavefronts per workgroup, on
    pictures 1 macroblock high / wide and ordinary ones; light (P) and heavy (intra) pictures both split"""
    tail(rows, rows, waves, rows, rows, waves, 1 << 20)
    rng = np.random.default_rng(seed)
    lib = built.lib()
    jobs = [build_job(lib, rng, wmb, hmb, 0, 4, [])]
    jobs.append(build_job(lib, rng, wmb, hmb, 1, 4, [0]))
    jobs.append(build_job(lib, rng, wmb, hmb, 2, 4, [0, 1], p_inter=0.9))
    jobs.append(build_job(lib, rng, wmb, hmb, 3, 4, [0, 1, 2], p_inter=0.97, mv_range=64))
    _run(built, jobs, n_streams=3)


def test_huge_picture_in_row_bands(built, tail):
    """4096x2304 (256 x 144 macroblocks, the largest level-5.1 frame): 8 bands of 18 rows (intra picture) and 4 bands of
    36 rows (P picture); the scheduling state of a band must fit the LDS next to its wavefronts"""
    tail(36, 18, 4, 36, 18, 4, 1 << 20)
    rng = np.random.default_rng(41)
    lib = built.lib()
    jobs = [build_job(lib, rng, 256, 144, 0, 2, []), build_job(lib, rng, 256, 144, 1, 2, [0], p_inter=0.9, mv_range=300)]
    _run(built, jobs, n_streams=1)


@pytest.mark.parametrize("name", ["damaged_41", "damaged_29", "damaged_32"])
def test_concealed_picture_next_to_a_banded_heavy_picture(built, tail, name):
    """one tick holds an I picture with concealed macroblocks (they may wait for the macroblock BELOW them:
    FjHeader.intra_down_deps, the picture must stay in ONE band of k_frame_intra) and an intact intra picture that wants
    a band per macroblock row.  The launch's rows-per-band cap used to follow the picture that wanted the most bands, so
    the concealed picture was split as well and read tiles below a band boundary before they were reconstructed."""
    import struct
    import h264bsd_amd
    from damage import damage
    from h264writer import StreamWriter
    from synth_configs import DAMAGED
    cfg, dmg = DAMAGED[name]
    jobs, _, _ = h264bsd_amd.capture_stream(damage(StreamWriter(**cfg).build(), **dmg))
    conc = jobs[0]
    h = pyoracle.blob_header(conc)
    assert struct.unpack_from("<I", bytes(conc), 96)[0] == 1 and h["n_intra_levels"] > 0, "fixture no longer has downward dependencies"
    wmb, hmb, n_slots = h["width_mbs"], h["height_mbs"], h["n_slots"]
    rng = np.random.default_rng(77)
    intact = build_job(built.lib(), rng, wmb, hmb, (h["cur_slot"] + 1) % n_slots, n_slots, [])
    want = {}
    for job in (conc, intact):
        want[pyoracle.blob_header(job)["cur_slot"]] = pyoracle.OracleDpb(job).decode(job)
    tail(1, 1, 4, 1, 1, 4, 1 << 20)                      # a band per macroblock row for whoever may be split
    for rep_no in range(4):
        rep = built.Replay([intact, conc], n_streams=2, offsets=[0, 1])
        try:
            rep.run(0, 2)                                # tick 0: stream 0 intact + stream 1 concealed; tick 1: the other way round
            for s in range(2):
                for slot, w in want.items():
                    got = rep.fetch(s, slot)
                    assert np.array_equal(got, w), f"run {rep_no} stream {s} slot {slot}: {np.count_nonzero(got != w)} bytes differ"
        finally:
            rep.close()
