#!/usr/bin/env python3
"""Model of a ROW-WORKER deblocking kernel against the ready-queue kernel (k_frame_dbk), on the real dependency graphs of the
bundled 1080p stream (flags as k_dbk computes them; tools/dbk_chains.py).

Today: a free wavefront claims up to eight ready macroblocks from a queue; a step is claim + one memory round trip + V pass +
H pass + stores + release, and a dependant can only be claimed after that.  Row workers: each 8-lane worker owns a macroblock
ROW and walks it left to right — the left strip stays in its registers / LDS tile, the next macroblock's tile and record are
requested a step ahead, the row above is awaited through a per-row progress counter in LDS — so a link of a dependency chain
costs the two filter passes and little else.  The price: the macroblocks of a row are done in order, and the eight workers of a
wavefront run in lock-step (a step lasts as long as its most expensive macroblock, waiting workers idle through it).
usage: dbk_rows_model.py [first picture] [count]"""
import os, sys, heapq
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import h264bsd_amd as h

data = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "test_1920x1080.h264"), "rb").read()
jobs, _, _ = h.capture_stream(data)
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else len(jobs)
Zx = [0, 1, 0, 1, 2, 3, 2, 3, 0, 1, 0, 1, 2, 3, 2, 3]; Zy = [0, 0, 1, 1, 0, 0, 1, 1, 2, 2, 3, 3, 2, 2, 3, 3]
zof = np.zeros((4, 4), int)
for z in range(16): zof[Zy[z], Zx[z]] = z
INTRA_KINDS = (1, 2, 3, 5, 6, 7)


def flags_of(j):
    hd = h.job_header(j); n, w, hh = hd["n_mbs"], hd["width_mbs"], hd["height_mbs"]
    rec = np.frombuffer(j, dtype=np.uint8, count=n * 32, offset=hd["rec_off"]).reshape(n, 32)
    kind, dbk = rec[:, 0], rec[:, 5]
    coded = np.frombuffer(rec[:, 8:12].tobytes(), dtype=np.uint32)
    refs = rec[:, 16:20]
    mv = h.job_mvs(j).reshape(n, 4, 4, 2)
    intra = np.isin(kind, INTRA_KINDS)
    cb = np.zeros((n, 4, 4), bool); rb = np.zeros((n, 4, 4), int)
    for y in range(4):
        for x in range(4):
            cb[:, y, x] = (coded >> zof[y, x]) & 1
            rb[:, y, x] = refs[:, (y >> 1) * 2 + (x >> 1)]
    g = lambda a: a.reshape(hh, w, 4, 4).transpose(0, 2, 1, 3).reshape(hh * 4, w * 4)
    CB, RB = g(cb), g(rb)
    MV = mv.reshape(hh, w, 4, 4, 2).transpose(0, 2, 1, 3, 4).reshape(hh * 4, w * 4, 2).astype(int)
    IN = np.repeat(np.repeat(intra.reshape(hh, w), 4, 0), 4, 1)
    H, W = hh * 4, w * 4
    def bs(p, q): return IN[p] | IN[q] | CB[p] | CB[q] | (RB[p] != RB[q]) | (np.abs(MV[p] - MV[q]).max(-1) >= 4)
    v = np.zeros((H, W), bool); v[:, 1:] = bs((slice(None), slice(0, W - 1)), (slice(None), slice(1, W)))
    hz = np.zeros((H, W), bool); hz[1:, :] = bs((slice(0, H - 1), slice(None)), (slice(1, H), slice(None)))
    D = dbk.reshape(hh, w)
    on = D != 0
    vb = v.reshape(hh, 4, w, 4); hb = hz.reshape(hh, 4, w, 4)
    left = vb[:, :, :, 0].any(axis=1) & ((D & 1) != 0)
    top = hb[:, 0, :, :].any(axis=2) & ((D & 2) != 0)
    inner = (vb[:, :, :, 1:].any(axis=(1, 3)) | hb[:, 1:, :, :].any(axis=(1, 3))) & on
    left &= on; top &= on
    return hh, w, left, top, inner, (left | top | inner)


def queue_model(hh, w, left, top, inner, any_, waves, per_mb, cost_full, cost_light):
    """k_frame_dbk as it is: two ready lists, up to per_mb macroblocks of one list per step"""
    n = hh * w
    dep = np.zeros(n, int); succ = [[] for _ in range(n)]
    A = any_.reshape(-1); Lf = left.reshape(-1); Tp = top.reshape(-1); In = inner.reshape(-1)
    for y in range(hh):
        for x in range(w):
            i = y * w + x
            if not A[i]: continue
            if x and Lf[i] and A[i - 1] and (In[i - 1] or Tp[i - 1]): dep[i] += 1; succ[i - 1].append(i)
            if y and Tp[i] and A[i - w] and (In[i - w] or Lf[i - w]): dep[i] += 1; succ[i - w].append(i)
            if y and x + 1 < w and Tp[i] and A[i - w + 1] and Lf[i - w + 1]: dep[i] += 1; succ[i - w + 1].append(i)
    ready = [[], []]
    def push(i): ready[1 if not In[i] else 0].append(i)
    for i in range(n):
        if A[i] and dep[i] == 0: push(i)
    t, free, running, done = 0, waves, [], 0
    total = int(A.sum())
    while done < total:
        while free and (ready[0] or ready[1]):
            q = 1 if len(ready[1]) >= len(ready[0]) and ready[1] else (0 if ready[0] else 1)
            batch, ready[q] = ready[q][:per_mb], ready[q][per_mb:]
            c = cost_light if q else cost_full
            heapq.heappush(running, (t + c, batch)); free -= 1
        t, batch = heapq.heappop(running)
        free += 1
        for i in batch:
            done += 1
            for s_ in succ[i]:
                dep[s_] -= 1
                if dep[s_] == 0: push(s_)
    return t


def row_model(hh, w, left, top, inner, any_, waves, cost_full, cost_light, poll, workers_per_wave=8, seg=0):
    """row workers.  seg > 0: a row is cut into segments of `seg` macroblocks, each its own task (the left strip then crosses
    workers at segment boundaries: it is awaited like the row above)."""
    A, Lf, Tp, In = any_, left, top, inner
    seg = seg or w
    # tasks in raster order of (row, segment): only those that hold a filtered macroblock
    tasks = [(y, s0) for y in range(hh) for s0 in range(0, w, seg) if A[y, s0:s0 + seg].any()]
    nxt = 0
    done_t = np.full((hh, w), -1.0)                         # time a macroblock was finished (unfiltered: 0)
    done_t[~A] = 0.0
    # a worker: (row, next column, end column) or None
    W = [[None] * workers_per_wave for _ in range(waves)]
    wave_t = [0.0] * waves
    finished = 0
    total = int(A.sum())
    def ready(y, x, t):
        f_top, f_left = Tp[y, x], Lf[y, x]
        if f_left and x and A[y, x - 1] and (In[y, x - 1] or Tp[y, x - 1]) and not (0 <= done_t[y, x - 1] <= t): return False
        if f_top and y:
            if A[y - 1, x] and (In[y - 1, x] or Lf[y - 1, x]) and not (0 <= done_t[y - 1, x] <= t): return False
            if x + 1 < w and A[y - 1, x + 1] and Lf[y - 1, x + 1] and not (0 <= done_t[y - 1, x + 1] <= t): return False
        return True
    ev = [(0.0, k) for k in range(waves)]
    heapq.heapify(ev)
    steps = active_sum = 0
    while finished < total:
        t, k = heapq.heappop(ev)
        ws = W[k]
        # idle workers take the next task
        for i in range(workers_per_wave):
            if ws[i] is None and nxt < len(tasks):
                y, s0 = tasks[nxt]; nxt += 1
                ws[i] = [y, s0, min(w, s0 + seg)]
        act = []
        for i in range(workers_per_wave):
            if ws[i] is None: continue
            y, x, xe = ws[i]
            while x < xe and not A[y, x]: x += 1            # unfiltered macroblocks are passed at once
            ws[i][1] = x
            if x >= xe: ws[i] = None; continue
            if ready(y, x, t): act.append(i)
        if not act:
            if all(v is None for v in ws) and nxt >= len(tasks): continue     # this wavefront is done
            heapq.heappush(ev, (t + poll, k)); continue
        c = cost_full if any(In[ws[i][0], ws[i][1]] for i in act) else cost_light
        for i in act:
            y, x, _ = ws[i]
            done_t[y, x] = t + c
            ws[i][1] = x + 1
            finished += 1
        steps += 1; active_sum += len(act)
        heapq.heappush(ev, (t + c, k))
    return max(done_t.max(), 0.0), steps, active_sum


GHZ = 2.4
F = [flags_of(jobs[i]) for i in range(first, min(first + count, len(jobs)))]
for label, fn in (
    ("ready queue, 8 wavefronts, 11500 / 4500 cycles per full / edge-only step (today)", lambda f: queue_model(*f, 8, 8, 11500, 4500)),
    ("ready queue, 8 wavefronts, 9500 / 3500 (stores off the critical path)", lambda f: queue_model(*f, 8, 8, 9500, 3500)),
    ("row workers, 8 wavefronts, 6500 / 3000, poll 400", lambda f: row_model(*f, 8, 6500, 3000, 400)[0]),
    ("row workers, 8 wavefronts, 7500 / 3500, poll 400", lambda f: row_model(*f, 8, 7500, 3500, 400)[0]),
    ("row workers, 8 wavefronts, 5500 / 2500, poll 400", lambda f: row_model(*f, 8, 5500, 2500, 400)[0]),
    ("row workers, 12 wavefronts, 6500 / 3000, poll 400", lambda f: row_model(*f, 12, 6500, 3000, 400)[0]),
    ("row workers, 4 wavefronts, 6500 / 3000, poll 400", lambda f: row_model(*f, 4, 6500, 3000, 400)[0]),
    ("row segments of 30, 8 wavefronts, 6500 / 3000", lambda f: row_model(*f, 8, 6500, 3000, 400, seg=30)[0]),
    ("row segments of 15, 8 wavefronts, 6500 / 3000", lambda f: row_model(*f, 8, 6500, 3000, 400, seg=15)[0]),
    ("row segments of 8, 8 wavefronts, 6500 / 3000", lambda f: row_model(*f, 8, 6500, 3000, 400, seg=8)[0]),
):
    cyc = sum(fn(f) for f in F)
    print(f"{label}: {cyc / GHZ / 1e6:.1f} ms per pass of {len(F)} pictures")

if len(sys.argv) > 3 and sys.argv[3] == "grid":
    for cf, cl in ((11500, 4500), (10500, 4000), (9500, 3500), (8500, 3200), (8000, 3000), (7000, 2500), (6000, 2200), (11500, 3000), (9500, 4500)):
        for waves in (8, 12):
            cyc = sum(queue_model(*f, waves, 8, cf, cl) for f in F)
            print(f"ready queue, {waves} wavefronts, {cf} / {cl}: {cyc / GHZ / 1e6:.1f} ms")


def queue_model4(hh, w, left, top, inner, any_, waves, per_mb, cost_full, cost_both, cost_one):
    """ready lists by class: inner / both macroblock edges / left edge only / upper edge only (a step of the last two runs ONE pass)"""
    n = hh * w
    dep = np.zeros(n, int); succ = [[] for _ in range(n)]
    A = any_.reshape(-1); Lf = left.reshape(-1); Tp = top.reshape(-1); In = inner.reshape(-1)
    for y in range(hh):
        for x in range(w):
            i = y * w + x
            if not A[i]: continue
            if x and Lf[i] and A[i - 1] and (In[i - 1] or Tp[i - 1]): dep[i] += 1; succ[i - 1].append(i)
            if y and Tp[i] and A[i - w] and (In[i - w] or Lf[i - w]): dep[i] += 1; succ[i - w].append(i)
            if y and x + 1 < w and Tp[i] and A[i - w + 1] and Lf[i - w + 1]: dep[i] += 1; succ[i - w + 1].append(i)
    ready = [[], [], [], []]
    cost = [cost_full, cost_both, cost_one, cost_one]
    def cls(i): return 0 if In[i] else 1 if (Lf[i] and Tp[i]) else 2 if Lf[i] else 3
    for i in range(n):
        if A[i] and dep[i] == 0: ready[cls(i)].append(i)
    t, free, running, done = 0, waves, [], 0
    total = int(A.sum())
    while done < total:
        while free and any(ready):
            q = max(range(4), key=lambda k: (len(ready[k]), k))
            batch, ready[q] = ready[q][:per_mb], ready[q][per_mb:]
            heapq.heappush(running, (t + cost[q], batch)); free -= 1
        t, batch = heapq.heappop(running)
        free += 1
        for i in batch:
            done += 1
            for s_ in succ[i]:
                dep[s_] -= 1
                if dep[s_] == 0: ready[cls(s_)].append(s_)
    return t


if len(sys.argv) > 3 and sys.argv[3] == "lists4":
    for cf, cb, co in ((11500, 4500, 4500), (11500, 4500, 3300), (11500, 4500, 3000), (11500, 4800, 3300)):
        cyc = sum(queue_model4(*f, 8, 8, cf, cb, co) for f in F)
        print(f"four ready lists, 8 wavefronts, inner {cf} / both edges {cb} / one edge {co}: {cyc / GHZ / 1e6:.1f} ms")
