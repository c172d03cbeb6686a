"""The frame job carries motion vectors compact (framejob.h): one vector in the record of a macroblock that has one, a sparse
64-byte entry for the others; the dense array is only the input of fj_finalize() and stays on the host.  CPU only."""
import os
import struct

import numpy as np
import pytest

import h264bsd_amd
from jobgen import build_job


@pytest.fixture(scope="module")
def lib():
    return h264bsd_amd.lib()


@pytest.mark.parametrize("seed", range(6))
def test_finished_job_reproduces_the_dense_vectors_it_was_built_from(lib, seed):
    rng = np.random.default_rng(8100 + seed)
    wmb, hmb = int(rng.integers(1, 9)), int(rng.integers(1, 7))
    seen = {}

    def keep(recs, mvs):
        # make some partitioned macroblocks uniform after all (sixteen equal vectors, one reference): fj_finalize must notice
        for a in range(recs.shape[0]):
            if recs[a, 0] == 0 and rng.random() < 0.3:
                mvs[a, :, :] = mvs[a, 0]
                recs[a, 16:20] = recs[a, 16]
        seen["recs"] = recs.copy(); seen["mvs"] = mvs.copy()

    blob = build_job(lib, rng, wmb, hmb, 3, 4, [0, 1, 2], p_inter=0.8, patch=keep)
    h = h264bsd_amd.job_header(blob)
    n = h["n_mbs"]
    rec = np.frombuffer(blob, dtype=np.uint8, count=n * 32, offset=h["rec_off"]).reshape(n, 32)
    inter = seen["recs"][:, 0] == 0
    got = h264bsd_amd.job_mvs(blob)
    assert np.array_equal(got[inter], seen["mvs"][inter])
    assert not got[~inter].any()
    # the flag of the finished job is exact: set where the sixteen vectors and the four references are equal, nowhere else
    uniform = np.array([inter[a] and (seen["mvs"][a] == seen["mvs"][a, 0]).all() and len(set(seen["recs"][a, 16:20])) == 1 for a in range(n)])
    assert np.array_equal((rec[:, 4] & 0x40) != 0, uniform)
    # sparse entries: one per inter macroblock with more than one vector, in address order, right behind each other
    many = np.nonzero(inter & ~uniform)[0]
    assert h["n_mvx"] == len(many)
    idx = np.frombuffer(rec[:, 28:32].tobytes(), dtype=np.uint32)
    assert np.array_equal(idx[many], np.arange(len(many)))
    assert h["mvx_off"] % 32 == 0 and h["total_bytes"] == ((h["mvx_off"] + 64 * len(many) + 31) & ~31)
    # partitioned list entries carry the same index (k_recon_inter<1> / <2> start from the entry, not from the record)
    for i in range(h["n_gen"]):
        mb, uni = struct.unpack_from("<HB", blob, h["gen_off"] + 16 * i)
        if uni != 1:
            assert struct.unpack_from("<I", blob, h["gen_off"] + 16 * i + 4)[0] == idx[mb]


def test_parser_jobs_keep_the_dense_array_off_the_wire():
    data = open(os.path.join(os.path.dirname(__file__), "golden", "test_640x360.h264"), "rb").read()
    jobs, _, _ = h264bsd_amd.capture_stream(data)
    assert jobs
    for j in jobs:
        h = h264bsd_amd.job_header(j)
        n = h["n_mbs"]
        assert len(j) == h["total_bytes"]
        assert h["mv_off"] >= h["total_bytes"]                      # the parser's dense array: past the end of what travels
        assert h["coef_off"] == h["rec_off"] + 32 * n                # nothing between records and coefficients any more
        rec = np.frombuffer(j, dtype=np.uint8, count=n * 32, offset=h["rec_off"]).reshape(n, 32)
        inter = rec[:, 0] == 0
        assert h["n_mvx"] == int((inter & ((rec[:, 4] & 0x40) == 0)).sum())
        mv = h264bsd_amd.job_mvs(j)
        skip_like = inter & ((rec[:, 4] & 0x40) != 0)
        assert (mv[skip_like] == mv[skip_like][:, :1, :]).all()
