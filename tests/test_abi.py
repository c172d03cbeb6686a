"""C-ABI surface: the library loads without a GPU and exports every symbol include/*.h declares."""
import ctypes
import os
import re
import subprocess

import pytest

from conftest import ROOT


def _declared_symbols(headers):
    names = set()
    for hdr in headers:
        text = open(os.path.join(ROOT, "include", hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(h264bsd\w+)\s*\(", text))
    names.discard("h264bsdmi_job_cb")
    return names


def _exported(path):
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return {line.split()[-1] for line in out.splitlines() if " T " in line}


def test_library_exports_every_declared_symbol(built):
    """the product library exports exactly the reference's API + the device extensions (no harness entry points); the
    bench library exports the harness of include/h264bsd_mi355x_bench.h on top"""
    product = _declared_symbols(("h264bsd_decoder.h", "h264bsd_mi355x.h"))
    everything = product | _declared_symbols(("h264bsd_mi355x_bench.h",))
    assert everything == set(built.EXPORTED_SYMBOLS)
    built.lib()
    assert _exported(built.LIB_PATH) == product
    assert _exported(built.capi.BENCH_LIB_PATH) == everything
    assert not any("Replay" in n or "Debug" in n for n in product)


def test_storage_t_keeps_reference_size():
    # reference sizeof(storage_t) on LP64 is 4648: callers put it on their stack
    text = open(os.path.join(ROOT, "include", "h264bsd_decoder.h")).read()
    m = re.search(r"unsigned char reserved\[(\d+)\]", text)
    assert ctypes.sizeof(ctypes.c_void_p) + int(m.group(1)) == 4648


def test_no_gpu_means_loud_failure_not_cpu_fallback(built, capfd):
    if built.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError):
        built.Decoder()                      # h264bsdInit must refuse: there is no CPU pixel path
    assert "no usable HIP device" in capfd.readouterr().err
    with pytest.raises(RuntimeError):
        built.convert(1, 16, 16, bytes(384))


def test_product_library_does_not_link_the_oracle(built):
    blob = open(built.LIB_PATH, "rb").read()
    assert b"oracle_" not in blob and b"liboracle" not in blob
