"""Damage a valid Annex-B stream at NAL level the way packet loss does: drop whole slice NAL units, cut slices
short, optionally flip a bit inside a slice (test infrastructure for tests/test_damaged_streams.py)."""
import numpy as np


def split_nals(data):
    idx, i = [], 0
    while True:
        j = data.find(b"\x00\x00\x00\x01", i)
        if j < 0:
            break
        idx.append(j)
        i = j + 4
    idx.append(len(data))
    return [data[idx[k]:idx[k + 1]] for k in range(len(idx) - 1)]


def damage(data, seed, p_drop=0.2, p_flip=0.0, p_trunc=0.2):
    r = np.random.default_rng(seed)
    out = bytearray()
    for k, n in enumerate(split_nals(data)):
        if (n[4] & 31) in (1, 5) and k > 3:          # slices only, never the parameter sets / first picture's first slice
            u = r.random()
            if u < p_drop:
                continue
            if u < p_drop + p_flip and len(n) > 12:
                b = bytearray(n)
                b[int(r.integers(8, len(n)))] ^= 1 << int(r.integers(0, 8))
                n = bytes(b)
            elif u < p_drop + p_flip + p_trunc and len(n) > 12:
                n = n[: int(r.integers(8, len(n)))]
        out += n
    return bytes(out)
