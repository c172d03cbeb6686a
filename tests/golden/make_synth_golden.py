#!/usr/bin/env python3
"""Generate tests/golden/synth_golden.json: for every synthetic stream of tests/test_synth_streams.py the sha1 of
the stream, the h264bsdDecode call trace and (sha1 of the frame, picId, isIdr, numErrMbs) of every output
picture in output order — all from the compiled REFERENCE decoder (oracle/_ref, built from /root/reference by
oracle/Makefile), run with every allocation starting out zeroed (synth.decode_reference) so that it is deterministic
also on the damaged streams listed in reference_undefined.json.  Run in the build container: python tests/golden/make_synth_golden.py"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import synth                      # noqa: E402
from synth_configs import CONFIGS, DAMAGED, DAMAGED_BUNDLED, FLIPPED, OVERFLOW, REDUNDANT, SWEEP_FINDS  # noqa: E402
from damage import damage  # noqa: E402
from h264writer import StreamWriter  # noqa: E402

out = {}
for name, cfg in CONFIGS.items():
    data = StreamWriter(**cfg).build()
    trace, pics = synth.decode_reference(data)
    assert not any(t[0] >= 3 for t in trace) and not any(p[3] for p in pics), f"{name}: the reference reports errors"
    assert len(pics) == cfg.get("n_pics", 6), name
    out[name] = dict(stream_sha1=hashlib.sha1(data).hexdigest(), bytes=len(data), trace=trace, pics=pics)
    print(name, len(data), "bytes", len(pics), "pictures")
for name, (cfg, dmg) in DAMAGED.items():
    data = damage(StreamWriter(**cfg).build(), **dmg)
    trace, pics = synth.decode_reference(data)
    out[name] = dict(stream_sha1=hashlib.sha1(data).hexdigest(), bytes=len(data), trace=trace, pics=pics)
    print(name, len(data), "bytes", len(pics), "pictures", sum(p[3] for p in pics), "concealed macroblocks")
undefined = {}
for name, (cfg, dmg) in list(FLIPPED.items()) + list(OVERFLOW.items()) + list(REDUNDANT.items()) + list(SWEEP_FINDS.items()):
    data = damage(StreamWriter(**cfg).build(), **dmg)
    if not synth.reference_is_deterministic(data):      # informational: the golden answers are those of the reference with zeroed frame buffers
        undefined[name] = "reference output changes with the heap fill byte (glibc M_PERTURB 0x55 / 0xAA): it shows or predicts from memory it never wrote"
    trace, pics = synth.decode_reference(data)
    out[name] = dict(stream_sha1=hashlib.sha1(data).hexdigest(), bytes=len(data), trace=trace, pics=pics)
    print(name, len(data), "bytes", len(pics), "pictures", sum(p[3] for p in pics), "concealed macroblocks",
          sum(1 for t in trace if t[0] == 3), "error calls")
json.dump(undefined, open(os.path.join(HERE, "reference_undefined.json"), "w"), indent=0, sort_keys=True)
for name, (stream, dmg) in DAMAGED_BUNDLED.items():
    data = damage(open(os.path.join(HERE, stream + ".h264"), "rb").read(), **dmg)
    trace, pics = synth.decode_reference(data)
    out[name] = dict(stream_sha1=hashlib.sha1(data).hexdigest(), bytes=len(data), trace=trace, pics=pics)
    print(name, len(data), "bytes", len(pics), "pictures", sum(p[3] for p in pics), "concealed macroblocks")
json.dump(out, open(os.path.join(HERE, "synth_golden.json"), "w"), indent=0)
