"""Sequence parameter sets with random contents — cropping, the whole VUI (aspect ratio, video signal type, chroma
location, timing, HRD, bitstream restriction), boundary and out-of-range values — through h264bsdDecode and the
information calls of the compiled reference (oracle/_ref) and of this library: the same accept / reject decisions
(reference src/h264bsd_seq_param_set.c, src/h264bsd_vui.c:95-372, DecodeHrdParameters :396-500), the same
h264bsdPicWidth / CroppingParams / VideoRange / MatrixCoefficients / SampleAspectRatio / Profile /
CheckValidParamSets, the same pictures.  The sweep itself is tools/sweep_headers.py (found: VUI range checks the parser
skipped, h264bsdCheckValidParamSets without the reference's CheckPps)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_random_sequence_parameter_sets_match_the_reference(built):
    from oracle import pyoracle
    if not os.path.exists(pyoracle.REF_SO):
        pytest.skip("oracle/_ref not built")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sweep_headers.py"), "0", "600"], capture_output=True, text=True, timeout=900)
    last = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ""
    assert r.returncode == 0 and "600 identical, 0 not" in last, r.stdout[-2000:]
    accepted = int(last.split("headers accepted in ")[1].split()[0])
    assert accepted > 60, "the generator no longer produces parameter sets the reference accepts"


@pytest.mark.parametrize("mode,count", [("pps", 300), ("slice", 200), ("nal", 200), ("bytestream", 200), ("multipps", 150), ("multisps", 60)])
def test_header_sweeps_match_the_reference(built, mode, count):
    """the other modes of tools/sweep_headers.py, a few hundred cases each: random picture parameter sets, random slice
    headers, skipped NAL unit types in mid-stream, Annex B framing variations, several PPSs / SPSs under different ids"""
    from oracle import pyoracle
    if not os.path.exists(pyoracle.REF_SO):
        pytest.skip("oracle/_ref not built")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sweep_headers.py"), "1000", str(count), mode],
                       capture_output=True, text=True, timeout=900)
    last = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ""
    assert r.returncode == 0 and f"{count} identical, 0 not" in last, r.stdout[-2000:]
