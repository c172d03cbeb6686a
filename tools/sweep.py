#!/usr/bin/env python3
"""Parity sweep on the CPU: random valid streams from the bitstream writer (tests/h264writer.py), optionally damaged,
decoded by the compiled reference (oracle/_ref, every allocation starting out zeroed: tests/synth.py) and by the host parser + oracle; the h264bsdDecode call traces and the
output pictures (hash, picId, isIdr, numErrMbs, order) must be identical.  TEST TOOL (uses oracle/): never imported by
the product.  usage: sweep.py <first seed> <count> [--damage] [--backend gpu]"""
import argparse, os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import h264writer, synth, damage as dmg

ap = argparse.ArgumentParser()
ap.add_argument("first", type=int); ap.add_argument("count", type=int)
ap.add_argument("--damage", action="store_true")
ap.add_argument("--flip", type=float, default=0.0, help="with --damage: probability that a slice NAL unit gets one flipped bit")
ap.add_argument("--drop", type=float, default=0.2); ap.add_argument("--trunc", type=float, default=0.2)
ap.add_argument("--overflow", type=float, default=0.0, help="probability that a coefficient block carries a level near the residual-range limit")
ap.add_argument("--keep-redundant", action="store_true"); ap.add_argument("--keep-gaps", action="store_true")
ap.add_argument("--huge-mv", type=float, default=0.0, help="share of motion vector differences of up to +-2300 samples (far outside the picture / out of range)")
ap.add_argument("--long", action="store_true", help="40-70 pictures per stream, pic_order_cnt_lsb of 4 bits, frame_num of 4: both wrap several times")
ap.add_argument("--sizes", default="", help="WMIN-WMAX,HMIN-HMAX: override the picture size of random_config (2-7 x 2-6 macroblocks) — tiny "
                "pictures, or rows longer than ten macroblocks (h264bsdMarkSliceCorrupted counts max(width, 10)); slice groups become none / dispersed")
ap.add_argument("--concat", type=int, default=1, help="N > 1: every case is N different random streams one after the other (new SPS "
                "/ PPS with the same ids, other picture and DPB sizes: re-activation, possibly in the middle of damage)")
ap.add_argument("--no-output-reordering", type=int, default=-1, help="h264bsdInit's flag: 0 / 1; default: seed parity for intact streams, 0 for damaged ones")
ap.add_argument("--still", type=float, default=0.0, help="writer options p_skip = this, p_intra_in_p = 0.02: long skip runs, i.e. static regions — where the product's copy elision leaves macroblocks out")
ap.add_argument("--backend", default="oracle", choices=("oracle", "gpu"), help="gpu = the product through the C ABI (needs an MI355X)")
args = ap.parse_args()
os.dup2(os.open(os.devnull, os.O_WRONLY), 2)      # the reference is built with _ERROR_PRINT
bad, t0, n_pics = [], time.time(), 0
for seed in range(args.first, args.first + args.count):
    try:
        parts = []
        for k in range(args.concat):
            sub = seed if args.concat == 1 else seed * args.concat + k
            cfg = h264writer.random_config(sub)
            if args.sizes:
                import random as _random
                (w0, w1), (h0, h1) = [tuple(int(v) for v in part.split("-")) for part in args.sizes.split(",")]
                rr = _random.Random(sub)
                cfg["wmb"], cfg["hmb"] = rr.randint(w0, w1), rr.randint(h0, h1)
                cfg["fmo"] = None if rr.random() < 0.6 or cfg["wmb"] * cfg["hmb"] < 2 else dict(type=1, groups=rr.randint(2, min(4, cfg["wmb"] * cfg["hmb"])))
                cfg["n_pics"] = min(cfg["n_pics"], 10)
            if args.damage:                         # as tests/synth_configs.py: no frame_num gaps, no redundant slices
                if not args.keep_gaps: cfg["gaps"] = 0
                if not args.keep_redundant: cfg["redundant"] = False
            if args.overflow: cfg["overflow"] = args.overflow; cfg["max_qp"] = max(cfg["max_qp"], 40)
            if args.huge_mv: cfg["p_huge_mv"] = args.huge_mv
            if args.still: cfg["p_skip"] = args.still; cfg["p_intra_in_p"] = 0.02
            if args.long:
                cfg["n_pics"] = 40 + sub % 31
            w = h264writer.StreamWriter(**cfg)
            if args.long:
                w.sps["log2_max_poc_lsb"] = 4
            part = w.build()
            if args.damage:
                part = dmg.damage(part, sub, p_drop=args.drop, p_flip=args.flip, p_trunc=args.trunc)
            if args.concat > 1 and k + 1 < args.concat and (seed + k) & 1:
                part = part[: len(part) * 2 // 3]       # the next sequence starts in the middle of this one
            parts.append(part)
        data = b"".join(parts)
        nor = args.no_output_reordering if args.no_output_reordering >= 0 else (seed & 1 if not args.damage else 0)
        ref = synth.decode_reference(data, nor)
        ours = synth.decode_ours(data, args.backend, nor)
        n_pics += len(ref[1])
        if (seed - args.first) % 100 == 99: print(f"... {seed - args.first + 1} streams, {len(bad)} not identical", flush=True)
        if ref != ours:
            bad.append(seed)
            print(f"MISMATCH seed {seed}: trace equal {ref[0] == ours[0]}, pictures {len(ref[1])} vs {len(ours[1])}", flush=True)
    except Exception as e:                      # a writer/config problem is reported, not hidden
        bad.append(seed)
        print(f"ERROR seed {seed}: {type(e).__name__}: {e}", flush=True)
print(f"[{args.backend}] seeds {args.first}..{args.first + args.count - 1}{' damaged' if args.damage else ''}: {args.count - len(bad)} identical, "
      f"{len(bad)} not ({bad[:20]}); {n_pics} pictures compared, {time.time() - t0:.0f} s")
