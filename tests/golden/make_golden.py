#!/usr/bin/env python3
"""Generate tests/golden/golden.json from the REAL reference decoder.

Run in the build container (needs oracle/_ref/libh264bsd_ref.so = /root/reference/src compiled by
`make -C oracle ref`).  The three .h264 files next to this script are the reference's own test data
(/root/reference/test/*.h264, copied as input fixtures).  For each stream the script records what the
reference does with it:
  * the h264bsdDecode call trace [(return code, readBytes)]
  * geometry / VUI getters
  * sha256 of every full, uncropped I420 output frame + sha256 of their concatenation
    (the latter equals SURVEY.md §8c / BASELINE.md)
  * the 64-bit position-weighted checksum (oracle.pyoracle.checksum64 == device kernel k_checksum)
  * sha256 of h264bsdConvertToRGBA/BGRA/YCbCrA of frames 0, 1 and 72
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle.pyoracle import RefDecoder, checksum64  # noqa: E402

STREAMS = ["test_640x360", "test_1920x1080", "test_1920x1080_fullRange"]


def main():
    ref = RefDecoder()
    out = {}
    for name in STREAMS:
        data = open(os.path.join(HERE, name + ".h264"), "rb").read()
        frames_sha, frames_ck, conv = [], [], {}
        total = hashlib.sha256()
        keep = {}

        def on_frame(f, frames_sha=frames_sha, frames_ck=frames_ck, total=total, keep=keep):
            i = len(frames_sha)
            frames_sha.append(hashlib.sha256(f.tobytes()).hexdigest())
            frames_ck.append(checksum64(f))
            total.update(f.tobytes())
            if i in (0, 1, 72):
                keep[i] = f.copy()

        trace, n, wmb, hmb = ref.decode_stream(data, on_frame)
        for i, f in keep.items():
            conv[str(i)] = [hashlib.sha256(ref.convert(fmt, wmb * 16, hmb * 16, f).tobytes()).hexdigest()
                            for fmt in range(3)]
        out[name] = dict(bytes=len(data), n_pics=n, width_mbs=wmb, height_mbs=hmb, trace=trace,
                         sha256_all=total.hexdigest(), frame_sha256=frames_sha, frame_checksum64=frames_ck,
                         convert_sha256=conv)
        print(name, n, wmb, hmb, total.hexdigest())
    json.dump(out, open(os.path.join(HERE, "golden.json"), "w"), indent=0)


if __name__ == "__main__":
    main()
