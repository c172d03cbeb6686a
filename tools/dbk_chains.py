#!/usr/bin/env python3
"""Dependency chains of the deblocking filter of the bundled 1080p stream, on the CPU: boundary strengths from the frame
jobs (8.7.2.1, as k_dbk computes them), the macroblock flags of k_frame_dbk (ANY / LEFT / TOP / INNER), and the longest
dependency path of every picture under
  (a) the rule k_frame_dbk uses (kernels.hip.h: a macroblock waits for (x-1,y), (x,y-1), (x+1,y-1) only where a sample
      they share can still change), one task per macroblock;
  (b) macroblock-granular dependencies without that refinement (every filtered neighbour counts);
  (c) TWO tasks per macroblock — V (vertical edges) and H (horizontal edges) — with the dependencies the samples impose:
      V(x,y) after H(x-1,y) if the left edge is active; H(x,y) after V(x,y), after H(x,y-1) if the upper edge is active and
      after V(x+1,y-1) if that macroblock's left edge is active (its V pass rewrites the columns of (x,y-1) that H(x,y)
      reads).  A path is then measured in half steps;
  (d) one task per macroblock as in (a), but a macroblock releases the one below-left of it after its V pass.
What a shorter path is worth: k_frame_dbk's P pictures are bound by links x ~12-15 k cycles.
usage: dbk_chains.py [first picture] [count]"""
import os, sys
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
INTRA_KINDS = (1, 2, 3, 5, 6, 7)      # framejob.h: everything that is filtered as intra (is_intra_kind)


def flags_of(j):
    hd = h.job_header(j); n, w, hh = hd["n_mbs"], hd["width_mbs"], hd["height_mbs"]
    rec = np.frombuffer(j, dtype=np.uint8, count=n * 32, offset=hd["rec_off"]).reshape(n, 32)
    kind, dbk, pred = rec[:, 0], rec[:, 5], rec[:, 4]
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
    v = np.zeros((H, W), bool); v[:, 1:] = bs((slice(None), slice(0, W - 1)), (slice(None), slice(1, W)))     # edge on the left of block x
    hz = np.zeros((H, W), bool); hz[1:, :] = bs((slice(0, H - 1), slice(None)), (slice(1, H), slice(None)))
    # motion is compared inside a macroblock only across the partition boundaries its type has: the writer of the jobs gives all
    # blocks of a partition one vector, so the comparison above says the same
    D = dbk.reshape(hh, w)
    on = D != 0
    vb = v.reshape(hh, 4, w, 4); hb = hz.reshape(hh, 4, w, 4)
    left = vb[:, :, :, 0].any(axis=1) & ((D & 1) != 0)            # FJ_DBK_LEFT = 1, FJ_DBK_TOP = 2 (framejob.h)
    top = hb[:, 0, :, :].any(axis=2) & ((D & 2) != 0)
    inner = (vb[:, :, :, 1:].any(axis=(1, 3)) | hb[:, 1:, :, :].any(axis=(1, 3))) & on
    left &= on; top &= on
    return hh, w, left, top, inner, (left | top | inner)


def longest(hh, w, left, top, inner, any_, rule):
    if rule == "d":
        # one task per macroblock as in (a), but (x+1,y-1) releases the macroblock below-left of it after its V pass: the
        # macroblock waits for the END of its left and upper neighbours and for the MIDDLE of the upper-right one
        L = np.zeros((hh, w), int)
        for y in range(hh):
            for x in range(w):
                if not any_[y, x]: continue
                d = 0
                if x and left[y, x] and (inner[y, x - 1] or top[y, x - 1]): d = max(d, L[y, x - 1])
                if y and top[y, x] and (inner[y - 1, x] or left[y - 1, x]): d = max(d, L[y - 1, x])
                if y and x + 1 < w and top[y, x] and left[y - 1, x + 1]: d = max(d, L[y - 1, x + 1] - 1)
                L[y, x] = d + 2
        return int(L.max())
    if rule in ("a", "b"):
        L = np.zeros((hh, w), int)
        for y in range(hh):
            for x in range(w):
                if not any_[y, x]: continue
                d = 0
                if rule == "a":
                    if x and left[y, x] and (inner[y, x - 1] or top[y, x - 1]): d = max(d, L[y, x - 1])
                    if y and top[y, x] and (inner[y - 1, x] or left[y - 1, x]): d = max(d, L[y - 1, x])
                    if y and x + 1 < w and top[y, x] and left[y - 1, x + 1]: d = max(d, L[y - 1, x + 1])
                else:
                    if x and any_[y, x - 1]: d = max(d, L[y, x - 1])
                    if y and any_[y - 1, x]: d = max(d, L[y - 1, x])
                    if y and x + 1 < w and any_[y - 1, x + 1]: d = max(d, L[y - 1, x + 1])
                L[y, x] = d + 2                                  # a macroblock = two half steps
        return int(L.max())
    V = np.zeros((hh, w), int); Hh = np.zeros((hh, w), int)
    for y in range(hh):
        for x in range(w):
            if not any_[y, x]: continue
            dv = Hh[y, x - 1] if (x and left[y, x] and any_[y, x - 1]) else 0
            V[y, x] = dv + 1
            dh = V[y, x]
            if y and top[y, x] and any_[y - 1, x]: dh = max(dh, Hh[y - 1, x])
            if y and x + 1 < w and top[y, x] and left[y - 1, x + 1]: dh = max(dh, V[y - 1, x + 1])
            Hh[y, x] = dh + 1
    return int(max(V.max(), Hh.max()))


tot = {"a": 0, "b": 0, "c": 0, "d": 0}
for i in range(first, min(first + count, len(jobs))):
    hh, w, left, top, inner, any_ = flags_of(jobs[i])
    r = {k: longest(hh, w, left, top, inner, any_, k) for k in ("a", "b", "c", "d")}
    for k in r: tot[k] += r[k]
    if count <= 12: print(f"picture {i}: filtered {int(any_.sum())}, half steps on the longest path: current rule {r['a']}, plain macroblock rule {r['b']}, V/H tasks {r['c']}")
print(f"pictures {first}..{first + count - 1}: half steps on the longest paths, summed: current rule {tot['a']} (= {tot['a'] // 2} links), plain macroblock rule {tot['b']}, V/H tasks {tot['c']}, early release of the macroblock below-left after the V pass {tot['d']}")


# ---- a model of k_frame_dbk's scheduler: W wavefronts, a free one takes up to 4 ready macroblocks (its share of the ready
# list when that is short), a step costs `step` cycles whatever it holds, dependants are released at its end ----
def simulate(hh, w, left, top, inner, any_, waves=12, step=12000, per_mb=4):
    import heapq
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
    ready = [i for i in range(n) if A[i] and dep[i] == 0]
    t, free, running, done, busy = 0, waves, [], 0, 0
    total = int(A.sum())
    while done < total:
        while free and ready:
            k = max(1, min(per_mb, len(ready) // max(1, free)))
            batch, ready = ready[:k], ready[k:]
            heapq.heappush(running, (t + step, batch)); free -= 1; busy += step
        t, batch = heapq.heappop(running)
        free += 1
        for i in batch:
            done += 1
            for s_ in succ[i]:
                dep[s_] -= 1
                if dep[s_] == 0: ready.append(s_)
    return t, busy


# ---- the same model with the step cost depending on what the step holds (round 4: eight macroblocks per step, and a question:
# what if macroblocks WITHOUT an active inner edge — they only need the first edge slot of each direction — were claimed apart
# from the others, so that their steps are short?) ----
def simulate2(hh, w, left, top, inner, any_, waves, per_mb, cost_full, cost_light, split):
    import heapq
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
    def push(i): ready[1 if (split and not In[i]) else 0].append(i)
    for i in range(n):
        if A[i] and dep[i] == 0: push(i)
    t, free, running, done, busy, steps = 0, waves, [], 0, 0, 0
    total = int(A.sum())
    while done < total:
        while free and (ready[0] or ready[1]):
            q = 1 if len(ready[1]) >= len(ready[0]) and ready[1] else (0 if ready[0] else 1)
            batch, ready[q] = ready[q][:per_mb], ready[q][per_mb:]
            light = all(not In[i] for i in batch)
            c = cost_light if light else cost_full
            heapq.heappush(running, (t + c, batch)); free -= 1; busy += c; steps += 1
        t, batch = heapq.heappop(running)
        free += 1
        for i in batch:
            done += 1
            for s_ in succ[i]:
                dep[s_] -= 1
                if dep[s_] == 0: push(s_)
    return t, busy, steps, total


# ---- chain following: a worker whose macroblock was the last thing (x+1, y) waited for goes on with that macroblock itself —
# no store wait, no release, no claim, no load of the strip between them (it is in the worker's LDS tile) ----
def simulate3(hh, w, left, top, inner, any_, waves, per_mb, cost_full, cost_light, follow_full, follow_light):
    import heapq
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
    t, free, running, done, busy, steps, followed = 0, waves, [], 0, 0, 0, 0
    total = int(A.sum())
    seq = 0
    while done < total:
        while free and (ready[0] or ready[1]):
            q = 1 if len(ready[1]) >= len(ready[0]) and ready[1] else (0 if ready[0] else 1)
            batch, ready[q] = ready[q][:per_mb], ready[q][per_mb:]
            c = cost_light if all(not In[i] for i in batch) else cost_full
            seq += 1; heapq.heappush(running, (t + c, seq, batch)); free -= 1; busy += c; steps += 1
        t, _, batch = heapq.heappop(running)
        nxt = []
        for i in batch:
            done += 1
            for s_ in succ[i]:
                dep[s_] -= 1
                if dep[s_] == 0:
                    if s_ == i + 1 and follow_full: nxt.append(s_)
                    else: push(s_)
        if nxt:
            c = follow_light if all(not In[i] for i in nxt) else follow_full
            seq += 1; heapq.heappush(running, (t + c, seq, nxt)); busy += c; steps += 1; followed += len(nxt)
        else: free += 1
    return t, busy, steps, total, followed


# ---- intra prediction and deblocking of a picture in ONE dataflow kernel?  Today k_frame_intra runs to its end before
# k_frame_dbk starts (a macroblock's filtering rewrites samples its neighbours' intra prediction still needs: 8.3 predicts from
# UNFILTERED samples), and in a P picture both are dependency chains that leave the device idle.  Fused, a macroblock's filtering
# waits for the intra macroblocks among itself and its eight neighbours only.  The model: W wavefronts; an intra step takes one
# macroblock (cost ci), a deblocking step up to 8 of one class (cost cf / cl); intra first. ----
def intra_graph(j):
    hd = h.job_header(j); n, w = hd["n_mbs"], hd["width_mbs"]
    rec = np.frombuffer(j, dtype=np.uint8, count=n * 32, offset=hd["rec_off"]).reshape(n, 32)
    idx = np.frombuffer(j, dtype=np.uint16, count=hd["n_intra"], offset=hd["idx_off"]).astype(int)
    sched = np.zeros(n, bool); sched[idx] = True
    need = rec[:, 16]
    D = [(-1, 0), (-1, -1), (0, -1), (1, -1), (1, 0), (1, 1), (0, 1), (-1, 1)]
    hh = n // w
    deps = {}
    for i in idx:
        x, y = i % w, i // w
        ds = []
        for b in range(8):
            if (need[i] >> b) & 1:
                nx, ny = x + D[b][0], y + D[b][1]
                if 0 <= nx < w and 0 <= ny < hh and sched[ny * w + nx]: ds.append(ny * w + nx)
        deps[int(i)] = ds
    return deps, sched


def simulate_fused(j, waves, ci, cf, cl, fused):
    import heapq
    hh, w, left, top, inner, any_ = flags_of(j)
    n = hh * w
    A = any_.reshape(-1); Lf = left.reshape(-1); Tp = top.reshape(-1); In = inner.reshape(-1)
    ideps, sched = intra_graph(j)
    # task ids: intra i -> i, deblock i -> n + i
    dep = {}; succ = {}
    def add(a, b): succ.setdefault(a, []).append(b); dep[b] = dep.get(b, 0) + 1
    for i, ds in ideps.items():
        dep.setdefault(i, 0)
        for d in ds: add(d, i)
    for y in range(hh):
        for x in range(w):
            i = y * w + x
            if not A[i]: continue
            t = n + i; dep.setdefault(t, 0)
            if x and Lf[i] and A[i - 1] and (In[i - 1] or Tp[i - 1]): add(n + i - 1, t)
            if y and Tp[i] and A[i - w] and (In[i - w] or Lf[i - w]): add(n + i - w, t)
            if y and x + 1 < w and Tp[i] and A[i - w + 1] and Lf[i - w + 1]: add(n + i - w + 1, t)
            if fused:
                for dy in (-1, 0, 1):
                    for dx in (-1, 0, 1):
                        nx, ny = x + dx, y + dy
                        if 0 <= nx < w and 0 <= ny < hh and sched[ny * w + nx]: add(ny * w + nx, t)
    def run(tasks_filter):
        d = {k: v for k, v in dep.items() if tasks_filter(k)}
        ready = [[], [], []]                      # intra, inner, edge-only
        def push(k): ready[0 if k < n else (1 if In[k - n] else 2)].append(k)
        for k, v in d.items():
            if v == 0: push(k)
        t, free, running, done, seq = 0, waves, [], 0, 0
        total = len(d)
        while done < total:
            while free and (ready[0] or ready[1] or ready[2]):
                if ready[0]: batch, ready[0] = ready[0][:1], ready[0][1:]; c = ci
                else:
                    q = 2 if len(ready[2]) >= len(ready[1]) and ready[2] else (1 if ready[1] else 2)
                    batch, ready[q] = ready[q][:8], ready[q][8:]; c = cl if q == 2 else cf
                seq += 1; heapq.heappush(running, (t + c, seq, batch)); free -= 1
            t, _, batch = heapq.heappop(running); free += 1
            for k in batch:
                done += 1
                for s_ in succ.get(k, []):
                    if s_ in d:
                        d[s_] -= 1
                        if d[s_] == 0: push(s_)
        return t
    if fused: return run(lambda k: True)
    # separate kernels: the deblocking graph without the cross edges, after the intra graph
    return run(lambda k: k < n) + run(lambda k: k >= n)



# ---- luma and chroma as TWO tasks per macroblock: the chroma filter of a macroblock depends on chroma samples only, the luma
# filter on luma samples only — two independent graphs of the same shape, whose steps are shorter than a joint one (the fixed
# part of a step — claim, one memory round trip, stores, release — is paid twice) ----
def simulate4(hh, w, left, top, inner, any_, waves, per_mb, fixed, pass_full, pass_light, luma_full, luma_light, split_planes):
    import heapq
    n = hh * w
    dep0 = np.zeros(n, int); succ = [[] for _ in range(n)]
    A = any_.reshape(-1); Lf = left.reshape(-1); Tp = top.reshape(-1); In = inner.reshape(-1)
    for y in range(hh):
        for x in range(w):
            i = y * w + x
            if not A[i]: continue
            if x and Lf[i] and A[i - 1] and (In[i - 1] or Tp[i - 1]): dep0[i] += 1; succ[i - 1].append(i)
            if y and Tp[i] and A[i - w] and (In[i - w] or Lf[i - w]): dep0[i] += 1; succ[i - w].append(i)
            if y and x + 1 < w and Tp[i] and A[i - w + 1] and Lf[i - w + 1]: dep0[i] += 1; succ[i - w + 1].append(i)
    parts = 2 if split_planes else 1
    dep = [dep0.copy() for _ in range(parts)]
    ready = [[] for _ in range(2 * parts)]          # [part][inner / edge-only]
    cost = {}
    if split_planes:
        cost = {0: fixed + pass_full * luma_full, 1: fixed + pass_light * luma_light,
                2: fixed + pass_full * (1 - luma_full), 3: fixed + pass_light * (1 - luma_light)}
    else:
        cost = {0: fixed + pass_full, 1: fixed + pass_light}
    def push(part, i): ready[2 * part + (0 if In[i] else 1)].append(i)
    for part in range(parts):
        for i in range(n):
            if A[i] and dep[part][i] == 0: push(part, i)
    t, free, running, done, busy, seq = 0, waves, [], 0, 0, 0
    total = int(A.sum()) * parts
    while done < total:
        while free and any(ready):
            # luma lists first (the long chain), the longer of the two; chroma when no luma is ready
            cand = [q for q in (0, 1) if ready[q]] or [q for q in range(2, 2 * parts) if ready[q]]
            q = max(cand, key=lambda q_: (len(ready[q_]), q_))
            batch, ready[q] = ready[q][:per_mb], ready[q][per_mb:]
            c = cost[q]
            seq += 1; heapq.heappush(running, (t + c, seq, q >> 1, batch)); free -= 1; busy += c
        t, _, part, batch = heapq.heappop(running)
        free += 1
        for i in batch:
            done += 1
            for s_ in succ[i]:
                dep[part][s_] -= 1
                if dep[part][s_] == 0: push(part, s_)
    return t, busy


if len(sys.argv) > 3 and sys.argv[3] == "simulate4":
    GHZ = 2.4
    for waves, fixed, pf, pl, lf, ll, sp in ((8, 3500, 7500, 3000, 0.82, 0.67, 0), (8, 3500, 7500, 3000, 0.82, 0.67, 1), (12, 3500, 7500, 3000, 0.82, 0.67, 1),
                                             (8, 3000, 6000, 2600, 0.82, 0.67, 0), (8, 3000, 6000, 2600, 0.82, 0.67, 1), (12, 3000, 6000, 2600, 0.82, 0.67, 1)):
        cyc = busy = 0
        for i in range(first, min(first + count, len(jobs))):
            hh, w, left, top, inner, any_ = flags_of(jobs[i])
            t, b = simulate4(hh, w, left, top, inner, any_, waves, 8, fixed, pf, pl, lf, ll, sp)
            cyc += t; busy += b
        print(f"model: {waves} wavefronts, fixed {fixed} + passes {pf} / {pl} cycles per step, {'luma and chroma apart' if sp else 'one task per macroblock'}: "
              f"{cyc / GHZ / 1e6:.1f} ms per pass, wavefronts busy {busy / (cyc * waves):.0%}")

if len(sys.argv) > 3 and sys.argv[3] == "fused":
    GHZ = 2.4
    for waves in (8, 12):
        sep = fus = 0
        for i in range(first, min(first + count, len(jobs))):
            sep += simulate_fused(jobs[i], waves, 9500, 11000, 6500, False)
            fus += simulate_fused(jobs[i], waves, 9500, 11000, 6500, True)
        print(f"model, {waves} wavefronts: k_frame_intra then k_frame_dbk {sep / GHZ / 1e6:.1f} ms per pass, one fused dataflow kernel {fus / GHZ / 1e6:.1f} ms")

if len(sys.argv) > 3 and sys.argv[3] == "simulate3":
    GHZ = 2.4
    for waves, per_mb, cf, cl, ff, fl in ((8, 8, 11000, 6500, 0, 0), (8, 8, 11000, 6500, 7500, 3500), (8, 8, 11000, 6500, 6500, 3000), (12, 8, 11000, 6500, 7500, 3500)):
        cyc = busy = steps = mbs = fol = 0
        for i in range(first, min(first + count, len(jobs))):
            hh, w, left, top, inner, any_ = flags_of(jobs[i])
            t, b, st, tot_, f_ = simulate3(hh, w, left, top, inner, any_, waves, per_mb, cf, cl, ff, fl)
            cyc += t; busy += b; steps += st; mbs += tot_; fol += f_
        print(f"model: {waves} wavefronts, up to {per_mb} per step, {cf} / {cl} cycles per full / edge-only step, a followed step {ff} / {fl}: "
              f"{cyc / GHZ / 1e6:.1f} ms per pass, {mbs / steps:.2f} macroblocks per step, {100.0 * fol / mbs:.0f} % of the macroblocks followed")

if len(sys.argv) > 3 and sys.argv[3] == "simulate2":
    GHZ = 2.4
    for waves, per_mb, cf, cl, split in ((8, 8, 11500, 11500, 0), (8, 8, 11500, 4500, 0), (8, 8, 11500, 4500, 1), (12, 8, 11500, 4500, 1), (8, 8, 11500, 6000, 1), (8, 8, 9000, 4500, 1),
                                         (8, 4, 8000, 8000, 0), (12, 4, 8000, 8000, 0), (12, 4, 8000, 3500, 1)):
        cyc = busy = steps = mbs = 0
        for i in range(first, min(first + count, len(jobs))):
            hh, w, left, top, inner, any_ = flags_of(jobs[i])
            t, b, st, tot_ = simulate2(hh, w, left, top, inner, any_, waves, per_mb, cf, cl, split)
            cyc += t; busy += b; steps += st; mbs += tot_
        print(f"model: {waves} wavefronts, up to {per_mb} per step, {cf} cycles per step ({cl} when no macroblock of it has an inner edge), {'two ready lists' if split else 'one ready list'}: "
              f"{cyc / GHZ / 1e6:.1f} ms per pass, wavefronts busy {busy / (cyc * waves):.0%}, {mbs / steps:.2f} macroblocks per step")

if len(sys.argv) > 3 and sys.argv[3] == "simulate":
    GHZ = 2.3
    for waves, step, per_mb in ((12, 12000, 4), (8, 12000, 4), (16, 12000, 4), (12, 12000, 2), (12, 9000, 4), (24, 12000, 4)):
        cyc = busy = 0
        for i in range(first, min(first + count, len(jobs))):
            hh, w, left, top, inner, any_ = flags_of(jobs[i])
            t, b = simulate(hh, w, left, top, inner, any_, waves, step, per_mb)
            cyc += t; busy += b
        print(f"model: {waves} wavefronts, {step} cycles per step, up to {per_mb} macroblocks per step: {cyc / GHZ / 1e6:.1f} ms per pass of these pictures, "
              f"wavefronts busy {busy / (cyc * waves):.0%}")
