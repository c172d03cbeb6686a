#!/usr/bin/env python3
"""Copy-elision invariant (tests/test_copy_elision.py: replay_with_elision) over random writer-made streams, intact or
damaged like tools/sweep.py makes them.  TEST TOOL (uses oracle/).  usage: sweep_elision.py <first seed> <count> [--damage]
[--flip P] [--drop P] [--trunc P] [--keep-redundant] [--keep-gaps] [--long] [--concat N]"""
import argparse, os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import h264writer, damage as dmg
import h264bsd_amd
from test_copy_elision import replay_with_elision

ap = argparse.ArgumentParser()
ap.add_argument("first", type=int); ap.add_argument("count", type=int)
ap.add_argument("--damage", action="store_true")
ap.add_argument("--flip", type=float, default=0.0); ap.add_argument("--drop", type=float, default=0.2); ap.add_argument("--trunc", type=float, default=0.2)
ap.add_argument("--keep-redundant", action="store_true"); ap.add_argument("--keep-gaps", action="store_true")
ap.add_argument("--long", action="store_true"); ap.add_argument("--concat", type=int, default=1)
ap.add_argument("--still", type=float, default=0.0, help="writer options p_skip = this, p_intra_in_p = 0.02: long skip runs, i.e. static regions, where elision happens")
args = ap.parse_args()
h264bsd_amd.build()
os.dup2(os.open(os.devnull, os.O_WRONLY), 2)
bad, t0, n_copy, n_elided = [], time.time(), 0, 0
for seed in range(args.first, args.first + args.count):
    parts = []
    for k in range(args.concat):
        sub = seed if args.concat == 1 else seed * args.concat + k
        cfg = h264writer.random_config(sub)
        if args.damage:
            if not args.keep_gaps: cfg["gaps"] = 0
            if not args.keep_redundant: cfg["redundant"] = False
        if args.long: cfg["n_pics"] = 40 + sub % 31
        if args.still: cfg["p_skip"] = args.still; cfg["p_intra_in_p"] = 0.02
        part = h264writer.StreamWriter(**cfg).build()
        if args.damage: part = dmg.damage(part, sub, p_drop=args.drop, p_flip=args.flip, p_trunc=args.trunc)
        parts.append(part)
    try:
        c, e = replay_with_elision(h264bsd_amd, b"".join(parts))
        n_copy += c; n_elided += e
    except AssertionError as ex:
        bad.append((seed, str(ex)[:200]))
    except Exception as ex:                                   # (e.g. a stream the parser refuses outright)
        bad.append((seed, "exception " + repr(ex)[:200]))
sys.stdout.write(f"{args.count} streams from seed {args.first}: copies {n_copy}, elided {n_elided}, {len(bad)} failures in {time.time() - t0:.0f} s\n")
for b in bad[:20]: sys.stdout.write(f"  {b}\n")
