"""The schedules of a frame job against its records.

The CPU oracle renders a job from the per-macroblock records alone; the kernels are launched over the lists that
fj_finalize() (h264bsd_amd/csrc/hd_core.c) derives from them: copy runs, general-inter entries, intra index with level
starts, deblocking index.  A list that disagrees with the records is invisible to every oracle-based CPU test, so this
one checks the lists themselves, on the error-path fixtures (ghost jobs, deblock-only jobs with second-phase
macroblocks, concealment): every macroblock that the oracle reconstructs appears in exactly one reconstruction list,
nothing else does, and the level starts partition the intra index."""
import ctypes
import struct

import pytest

import h264bsd_amd
from h264bsd_amd import capi
from synth_configs import DAMAGED, FLIPPED, REDUNDANT
from test_damaged_streams import UNDEFINED, stream_of

ABSENT, STALE, INTER, CONCEAL_P, PHASE2 = 255, 6, 0, 5, 0x80
NAMES = [n for n in list(REDUNDANT)[:24] + list(FLIPPED)[:12] + list(DAMAGED)[:12] if n not in UNDEFINED]


def jobs_of(name):
    data, jobs = stream_of(name), []
    dec = capi.Decoder(0, capture=lambda b: jobs.append(bytes(b)))
    buf = ctypes.create_string_buffer(data, len(data))
    base, off, pid, stall = ctypes.addressof(buf), 0, 0, 0
    while off < len(data) and stall <= 3:
        r, rb = dec.decode(base + off, len(data) - off, pid)
        off += rb
        pid += r == 1
        stall = stall + 1 if rb == 0 else 0
    dec.close()
    return jobs


@pytest.mark.parametrize("name", NAMES)
def test_lists_cover_exactly_what_the_records_reconstruct(built, name):
    for j in jobs_of(name):
        h = h264bsd_amd.job_header(j)
        n = h["n_mbs"]
        kind = [j[h["rec_off"] + 32 * a] for a in range(n)]
        pred = [j[h["rec_off"] + 32 * a + 4] for a in range(n)]
        recon = {a for a in range(n) if kind[a] not in (ABSENT, STALE) and (not h["dbk_only"] or pred[a] & PHASE2)}
        copied = []
        for i in range(h["n_copy"]):
            mb, _slot, count, _dx, _dy = struct.unpack_from("<HBBhh", j, h["copy_off"] + 8 * i)
            copied += list(range(mb, mb + count))
        gen = [struct.unpack_from("<H", j, h["gen_off"] + 16 * i)[0] for i in range(h["n_gen"])]
        intra = [struct.unpack_from("<H", j, h["idx_off"] + 2 * i)[0] for i in range(h["n_intra"])]
        listed = copied + gen + intra
        assert len(listed) == len(set(listed)), "a macroblock is scheduled twice"
        assert set(listed) == recon, (sorted(set(listed) ^ recon), h["ghost"], h["dbk_only"])
        assert all(kind[a] in (INTER, CONCEAL_P) for a in copied + gen)
        assert h["n_gen_uniform"] + h["n_gen_quad"] <= h["n_gen"]
        # the three parts of the general-inter list (k_recon_inter<0/1/2> each walk one): one motion vector per macroblock,
        # one per 8x8 quadrant, finer partitions
        kinds = [j[h["gen_off"] + 16 * i + 2] for i in range(h["n_gen"])]
        assert kinds == [1] * h["n_gen_uniform"] + [2] * h["n_gen_quad"] + [0] * (h["n_gen"] - h["n_gen_uniform"] - h["n_gen_quad"])
        lvl = struct.unpack_from(f"<{h['n_intra_levels'] + 1}I", j, h["lvl_off"]) if h["n_intra"] else (0,)
        assert list(lvl) == sorted(lvl) and lvl[0] == 0 and lvl[-1] == h["n_intra"]
        dbk = [struct.unpack_from("<H", j, h["dbk_off"] + 2 * i)[0] for i in range(h["n_dbk"])]
        assert len(dbk) == len(set(dbk)) and all(a < n for a in dbk)
        if h["ghost"]:
            assert not dbk and not h["any_deblock"]
