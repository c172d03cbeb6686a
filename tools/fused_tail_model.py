#!/usr/bin/env python3
"""Critical paths of the per-picture phase of the bundled 1080p stream as TWO dependency graphs walked one after the other
(k_frame_intra, then k_frame_dbk: what the product does) and as ONE graph in which the filter of a macroblock waits only for the
intra macroblocks of its 3 x 3 neighbourhood (tools/experiments/r6_fused_tail.patch: k_frame_tail).  Unbounded workers, 5 us per
intra link, 4 us per filter link; flags and rules from tools/dbk_chains.py.  The model promised 32.4 -> 26.1 ms per step for the
71 P pictures; the kernel that was built did not deliver it (docs/EXPERIMENTS.md): with twelve wavefronts a heavy P picture's
phase is bound by what the compute unit can issue, not by the paths.  usage: fused_tail_model.py"""
import os, sys, runpy
import numpy as np
sys.argv = ["dbk_chains.py", "0", "0"]
g = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "dbk_chains.py"))
flags_of, jobs, h = g["flags_of"], g["jobs"], g["h"]
WI, WD = 5.0, 4.0      # us per link: intra, dbk
tot_sep = tot_fused = tot_sep_p = tot_fused_p = 0.0
rows = []
for i, j in enumerate(jobs):
    hd = h.job_header(j); n, w, hh = hd["n_mbs"], hd["width_mbs"], hd["height_mbs"]
    rec = np.frombuffer(j, dtype=np.uint8, count=n * 32, offset=hd["rec_off"]).reshape(n, 32)
    _, _, left, top, inner, anyf = flags_of(j)
    # intra schedule: level per MB from the record (intra_level u16 where?) -> recompute from the idx list order is hard; use header levels and the need masks
    idx = np.frombuffer(j, dtype=np.uint16, count=hd["n_intra"], offset=hd["idx_off"]).astype(int)
    lvl = np.frombuffer(j, dtype=np.uint32, count=hd["n_intra_levels"] + 1, offset=hd["lvl_off"]).astype(int) if hd["n_intra_levels"] else np.zeros(1, int)
    ilevel = -np.ones(n, int)
    for l in range(hd["n_intra_levels"]):
        ilevel[idx[lvl[l]:lvl[l + 1]]] = l
    t_intra = np.where(ilevel >= 0, (ilevel + 1) * WI, 0.0).reshape(hh, w)      # finish time of intra MB (dataflow from t=0)
    # dbk separate: start at 0 relative to its kernel
    def dbk_path(ready):
        fin = np.zeros((hh, w))
        for y in range(hh):
            for x in range(w):
                if not anyf[y, x]: continue
                s = ready[y, x]
                if left[y, x] and x > 0 and anyf[y, x - 1] and (inner[y, x - 1] or top[y, x - 1]): s = max(s, fin[y, x - 1])
                if top[y, x] and y > 0:
                    if anyf[y - 1, x] and (inner[y - 1, x] or left[y - 1, x]): s = max(s, fin[y - 1, x])
                    if x + 1 < w and anyf[y - 1, x + 1] and left[y - 1, x + 1]: s = max(s, fin[y - 1, x + 1])
                fin[y, x] = s + WD
        return fin.max()
    sep = (hd["n_intra_levels"] * WI) + dbk_path(np.zeros((hh, w)))
    # fused: dbk(m) ready when every intra MB in its 3x3 neighbourhood is done
    pad = np.zeros((hh + 2, w + 2)); pad[1:-1, 1:-1] = t_intra
    ready = np.zeros((hh, w))
    for dy in range(3):
        for dx in range(3):
            ready = np.maximum(ready, pad[dy:dy + hh, dx:dx + w])
    fused = max(dbk_path(ready), t_intra.max())
    rows.append((i, hd["n_intra"], hd["n_intra_levels"], sep, fused))
    tot_sep += sep; tot_fused += fused
    if hd["n_intra"] < 4000: tot_sep_p += sep; tot_fused_p += fused
for r in rows: print("pic %2d n_intra %4d levels %3d  separate %7.1f us  fused %7.1f us" % r)
print("all: separate %.1f ms fused %.1f ms; P pictures: %.1f -> %.1f ms" % (tot_sep / 1e3, tot_fused / 1e3, tot_sep_p / 1e3, tot_fused_p / 1e3))
