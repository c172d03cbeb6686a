"""A compiled C application of the drop-in library ("the boundary is exercised through ctypes only").

tests/c_caller/decode_hash.c is shaped like the reference's posix/test_h264bsd.c:127-183 — storage_t on the caller's
stack, one h264bsdDecode() per NAL unit, pictures pulled after PIC_RDY — compiled against include/h264bsd_decoder.h and
linked with -lh264bsd_mi355x.  It prints the SHA-256 of the concatenated frames; the expected values are the known
answers of the compiled reference (tests/golden/golden.json "sha256_all").

CPU: the program builds and links against the product library (every symbol it needs is exported with the reference's
signature), and the SAME source linked against the reference library instead prints the golden hashes (which pins the
program itself).  GPU: the product-linked binary prints them — single decoder, `-r 2` (re-initialising a stack
storage_t), and 8 decoders on 8 threads."""
import os
import subprocess

import pytest

from conftest import GOLDEN_DIR, ROOT, STREAMS
from oracle import pyoracle

SRC = os.path.join(ROOT, "tests", "c_caller", "decode_hash.c")
LIBDIR = os.path.join(ROOT, "h264bsd_amd", "lib")


def _build(tmp, libdir, lib):
    exe = str(tmp / f"decode_hash_{lib}")
    subprocess.run(["gcc", "-O2", "-Wall", "-Wextra", "-Werror", "-std=gnu11", "-I" + os.path.join(ROOT, "include"), SRC, "-o", exe,
                    "-L" + libdir, "-l" + lib, "-lpthread", "-Wl,-rpath," + libdir], check=True)
    return exe


def _digests(exe, args, stream, env=None):
    out = subprocess.run([exe, *args, os.path.join(GOLDEN_DIR, stream + ".h264")], capture_output=True, text=True, timeout=900,
                         env=dict(os.environ, **env) if env else None)
    assert out.returncode == 0, out.stderr[-800:]
    return [(int(line.split()[4]), line.split()[-1]) for line in out.stdout.splitlines() if line.startswith("decoder")]


def test_c_caller_links_against_the_product_and_is_pinned_by_the_reference(tmp_path, built, golden):
    _build(tmp_path, LIBDIR, "h264bsd_mi355x")                      # link check: the reference's symbols and signatures
    if not os.path.exists(pyoracle.REF_SO):
        pytest.skip("oracle/_ref not built")
    exe = _build(tmp_path, os.path.dirname(pyoracle.REF_SO), "h264bsd_ref")
    assert _digests(exe, [], "test_640x360") == [(73, golden["test_640x360"]["sha256_all"])]


@pytest.mark.gpu
@pytest.mark.parametrize("stream", STREAMS)
def test_gpu_c_caller_decodes_the_bundled_streams(tmp_path, built, golden, stream):
    exe = _build(tmp_path, LIBDIR, "h264bsd_mi355x")
    assert _digests(exe, [], stream) == [(73, golden[stream]["sha256_all"])]


@pytest.mark.gpu
def test_gpu_c_caller_repeat_and_threads(tmp_path, built, golden):
    exe = _build(tmp_path, LIBDIR, "h264bsd_mi355x")
    want = (73, golden["test_640x360"]["sha256_all"])
    assert _digests(exe, ["-r", "2"], "test_640x360") == [want, want]
    assert sorted(_digests(exe, ["-t", "8"], "test_640x360")) == [want] * 8
    assert sorted(_digests(exe, ["-t", "4", "-r", "2"], "test_1920x1080")) == [(73, golden["test_1920x1080"]["sha256_all"])] * 8


@pytest.mark.gpu
def test_gpu_c_caller_threads_under_uneven_timing(tmp_path, built, golden):
    """8 decoders on 8 threads, repeated, each thread also writing its frames to a file (DH_DUMP): the threads drift apart,
    the engine's ticks mix I and P pictures of different instances on its lanes, and the per-picture kernels run as row bands
    next to each other.  (Found that way: the ticket counter of a lane's first banded launch was zeroed by a hipMemset() that
    had not run yet when the launch started — one run in seven printed a wrong digest.)"""
    exe = _build(tmp_path, LIBDIR, "h264bsd_mi355x")
    want = (73, golden["test_640x360"]["sha256_all"])
    dump = tmp_path / "frames"
    dump.mkdir()
    for run in range(10):
        assert sorted(_digests(exe, ["-t", "8"], "test_640x360", env={"DH_DUMP": str(dump)})) == [want] * 8, f"run {run}"
