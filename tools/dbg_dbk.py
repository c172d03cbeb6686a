#!/usr/bin/env python3
"""Debug aid: random hand-built pictures through the kernels and the oracle, differences shown per macroblock (which rows /
columns of which plane).  tools/dbg_dbk.py [wmb hmb seed n_pics stages]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import h264bsd_amd
from oracle import pyoracle
from jobgen import build_job

def main():
    wmb, hmb, seed, n_pics, stages = (int(x) for x in (sys.argv[1:6] + ['3', '2', '1', '2', '7'][len(sys.argv) - 1:]))
    rng = np.random.default_rng(seed)
    lib = h264bsd_amd.lib()
    jobs = [build_job(lib, rng, wmb, hmb, 0, 4, [])]
    for i in range(1, n_pics): jobs.append(build_job(lib, rng, wmb, hmb, i, 4, list(range(i)), p_inter=0.8))
    rep = h264bsd_amd.Replay(jobs, n_streams=1)
    rep.set_stages(stages)
    dpb = pyoracle.OracleDpb(jobs[0])
    W, H = wmb * 16, hmb * 16
    for i, job in enumerate(jobs):
        rep.run(i, 1)
        want = dpb.decode(job, deblock=bool(stages & 4))
        got = rep.fetch(0, pyoracle.blob_header(job)['cur_slot'])
        print(f"picture {i}: {np.count_nonzero(got != want)} bytes differ")
        gy, wy = got[:W * H].reshape(H, W), want[:W * H].reshape(H, W)
        gc, wc = got[W * H:].reshape(2, H // 2, W // 2), want[W * H:].reshape(2, H // 2, W // 2)
        for my in range(hmb):
            for mx in range(wmb):
                d = gy[16 * my:16 * my + 16, 16 * mx:16 * mx + 16] != wy[16 * my:16 * my + 16, 16 * mx:16 * mx + 16]
                if d.any():
                    print(f"  MB ({mx},{my}) luma: rows {sorted(set(np.nonzero(d)[0].tolist()))} cols {sorted(set(np.nonzero(d)[1].tolist()))}")
                    r, c = np.argwhere(d)[0]
                    print(f"     first ({c},{r}): got {gy[16 * my + r, 16 * mx + c]} want {wy[16 * my + r, 16 * mx + c]}; row got {gy[16 * my + r, 16 * mx:16 * mx + 16].tolist()} want {wy[16 * my + r, 16 * mx:16 * mx + 16].tolist()}")
                for p in range(2):
                    d = gc[p, 8 * my:8 * my + 8, 8 * mx:8 * mx + 8] != wc[p, 8 * my:8 * my + 8, 8 * mx:8 * mx + 8]
                    if d.any():
                        print(f"  MB ({mx},{my}) chroma {p}: rows {sorted(set(np.nonzero(d)[0].tolist()))} cols {sorted(set(np.nonzero(d)[1].tolist()))}")
    rep.close()

if __name__ == '__main__':
    main()
