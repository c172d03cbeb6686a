#!/usr/bin/env python3
"""What the general-inter list of a stream asks of k_recon_inter: interpolation classes (8.4.2.2.1 letters), coded / uncoded
macroblocks, coded 4x4 blocks per macroblock.  tools/gen_stats.py [stream.h264]"""
import sys, os, collections
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import h264bsd_amd

LETTER = {(0, 0): 'G', (1, 0): 'a', (2, 0): 'b', (3, 0): 'c', (0, 1): 'd', (0, 2): 'h', (0, 3): 'n', (1, 1): 'e', (3, 1): 'g', (1, 3): 'p',
          (3, 3): 'r', (2, 2): 'j', (2, 1): 'f', (2, 3): 'q', (1, 2): 'i', (3, 2): 'k'}
GROUP = {'G': 'whole', 'a': 'hor', 'b': 'hor', 'c': 'hor', 'd': 'ver', 'h': 'ver', 'n': 'ver', 'e': 'diag', 'g': 'diag', 'p': 'diag', 'r': 'diag',
         'j': 'centre', 'f': 'centre', 'q': 'centre', 'i': 'centre', 'k': 'centre'}
gen_dt = np.dtype([('mb', '<u2'), ('uniform', 'u1'), ('slot', 'u1'), ('mvx', '<i2'), ('mvy', '<i2'), ('coef_idx', '<u4'), ('coded', '<u4')])

def main():
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'tests', 'golden', 'test_1920x1080.h264')
    jobs, _, _ = h264bsd_amd.capture_stream(open(path, 'rb').read(), copy_elision=True)
    letters = collections.Counter(); groups = collections.Counter(); chroma_frac = collections.Counter()
    n_uni = n_quad = n_rest = 0
    coded_luma = coded_chroma = coded_any = 0
    nblk = collections.Counter()
    n_mbs = 0
    for job in jobs:
        b = bytes(job)
        h = h264bsd_amd.job_header(b)
        n_mbs += h['n_mbs']
        g = np.frombuffer(b, dtype=gen_dt, count=h['n_gen'], offset=h['gen_off'])
        u = g[:h['n_gen_uniform']]
        n_uni += len(u); n_quad += h['n_gen_quad']; n_rest += h['n_gen'] - h['n_gen_uniform'] - h['n_gen_quad']
        for fx, fy in zip(u['mvx'] & 3, u['mvy'] & 3):
            l = LETTER[(int(fx), int(fy))]; letters[l] += 1; groups[GROUP[l]] += 1
        for fx, fy in zip(u['mvx'] & 7, u['mvy'] & 7):
            chroma_frac['whole' if (fx | fy) == 0 else 'one-dim' if fx == 0 or fy == 0 else 'two-dim'] += 1
        c = g['coded']
        coded_luma += int(np.count_nonzero(c & 0x0100FFFF)); coded_chroma += int(np.count_nonzero(c & 0x02FF0000)); coded_any += int(np.count_nonzero(c & 0x03FFFFFF))
        for v in c: nblk[bin(int(v) & 0xFFFFFF).count('1')] += 1
    tot = n_uni + n_quad + n_rest
    print(f"{len(jobs)} pictures, {n_mbs} macroblocks; general-inter list {tot} ({100.0 * tot / n_mbs:.1f} %): uniform {n_uni}, quadrant {n_quad}, finer {n_rest}")
    print(f"coded: any {coded_any} ({100.0 * coded_any / tot:.1f} %), luma {coded_luma} ({100.0 * coded_luma / tot:.1f} %), chroma {coded_chroma} ({100.0 * coded_chroma / tot:.1f} %)")
    print("uniform entries by luma interpolation class:", ', '.join(f"{k} {v} ({100.0 * v / n_uni:.1f} %)" for k, v in groups.most_common()))
    print("  letters:", ', '.join(f"{k} {100.0 * v / n_uni:.1f}" for k, v in sorted(letters.items())))
    print("uniform entries by chroma fraction:", ', '.join(f"{k} {100.0 * v / n_uni:.1f} %" for k, v in chroma_frac.most_common()))
    print("coded 4x4 blocks (luma + chroma AC) per entry:", ', '.join(f"{k}: {100.0 * v / tot:.1f} %" for k, v in sorted(nblk.items())[:12]))

if __name__ == '__main__':
    main()
