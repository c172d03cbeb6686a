"""The parser's fast paths (hd_mb.c: decode_skip_fast) against its general path: HD_NO_FAST_SKIP=1 in the environment makes
every macroblock take decode_mb_body.  Both must produce the same frame jobs, byte for byte, and the same call trace — on the
bundled, the synthetic and the damaged streams (where the fast path must decline whenever an error path could be reached)."""
import hashlib
import json
import os
import subprocess
import sys

from conftest import ROOT

WORKER = r"""
import sys, os, json, hashlib
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import h264bsd_amd
from conftest import STREAMS, stream_bytes
from h264writer import StreamWriter
from synth_configs import CONFIGS
import test_damaged_streams as dmg
import pytest
out = {}
def digest(data):
    jobs, trace, info = h264bsd_amd.capture_stream(data)
    h = hashlib.sha256()
    for j in jobs: h.update(j)
    h.update(json.dumps(trace).encode())
    return h.hexdigest(), len(jobs)
for name in STREAMS: out[name] = digest(stream_bytes(name))
for name, cfg in CONFIGS.items(): out["synth_" + name] = digest(StreamWriter(**cfg).build())
for name in dmg.NAMES:
    try: data = dmg.stream_of(name)
    except pytest.skip.Exception: continue
    out["dmg_" + name] = digest(data)
print(json.dumps(out))
""" % (ROOT, ROOT)


def _run(env_extra):
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, "-c", WORKER], capture_output=True, text=True, timeout=1500, env=env, cwd=os.path.join(ROOT, "tests"))
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def test_fast_skip_path_writes_what_the_general_path_writes(built):
    fast, general = _run({"HD_NO_FAST_SKIP": "0"}), _run({"HD_NO_FAST_SKIP": "1"})
    assert fast.keys() == general.keys() and len(fast) > 250
    differ = [k for k in fast if fast[k] != general[k]]
    assert not differ, f"frame jobs / call traces differ between the fast and the general path: {differ[:10]}"
