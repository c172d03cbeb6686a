"""C-ABI surface: the library loads without a GPU and exports every symbol include/*.h declares."""
import ctypes
import os
import re
import subprocess
import sys

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


# ---- the kernels of the built library were not compiled with -mtgsplit ----
def _device_code_object(path):
    """the gfx950 code object inside a host object / shared library (clang offload bundle, parsed by hand)"""
    import struct
    blob = open(path, "rb").read()
    i = blob.find(b"__CLANG_OFFLOAD_BUNDLE__")
    assert i >= 0, f"{path}: no offload bundle"
    n = struct.unpack_from("<Q", blob, i + 24)[0]
    off = i + 32
    for _ in range(n):
        o, sz, tl = struct.unpack_from("<QQQ", blob, off)
        off += 24
        triple = blob[off:off + tl].decode()
        off += tl
        if "gfx950" in triple:
            return blob[i + o:i + o + sz]
    raise AssertionError(f"{path}: no gfx950 code object")


def _kernel_descriptors(co):
    """name -> the 64-byte kernel descriptor (symbols <kernel>.kd of an AMDGPU ELF)"""
    import struct
    assert co[:4] == b"\x7fELF"
    shoff, = struct.unpack_from("<Q", co, 0x28)
    shentsize, shnum, _ = struct.unpack_from("<HHH", co, 0x3A)
    secs = [struct.unpack_from("<IIQQQQIIQQ", co, shoff + k * shentsize) for k in range(shnum)]
    out = {}
    for s in secs:
        if s[1] not in (2, 11):                           # SHT_SYMTAB / SHT_DYNSYM
            continue
        strtab = secs[s[6]]
        for k in range(s[5] // 24):
            name_off, _info, _other, shndx, value, size = struct.unpack_from("<IBBHQQ", co, s[4] + k * 24)
            end = co.index(b"\0", strtab[4] + name_off)
            name = co[strtab[4] + name_off:end].decode()
            if name.endswith(".kd") and size == 64 and 0 < shndx < shnum:
                sec = secs[shndx]
                out[name[:-3]] = co[sec[4] + value - sec[3]:sec[4] + value - sec[3] + 64]
    return out


def _tg_split(kd):
    import struct
    return (struct.unpack_from("<I", kd, 44)[0] >> 16) & 1      # COMPUTE_PGM_RSRC3.TG_SPLIT (gfx90a / gfx94x / gfx950)


def test_kernels_are_not_built_with_tgsplit(built, tmp_path):
    """release_stores() (kernels.hip.h) hands macroblocks over inside a workgroup without waiting for the stores: that is only
    right while a workgroup lives on ONE compute unit.  -mtgsplit defines no macro, so the built kernels are looked at: the
    threadgroup-split bit of every kernel descriptor must be clear.  The checker is validated on a two-line kernel first."""
    src = tmp_path / "k.hip"
    src.write_text("#include <hip/hip_runtime.h>\n__global__ void kk(int *p) { p[threadIdx.x] = 1; }\n")
    for flag, want in (([], 0), (["-mtgsplit"], 1)):
        obj = tmp_path / f"k{want}.o"
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-c", str(src), "-o", str(obj)] + flag, check=True, capture_output=True)
        kds = _kernel_descriptors(_device_code_object(str(obj)))
        assert kds and all(_tg_split(kd) == want for kd in kds.values())
    kds = _kernel_descriptors(_device_code_object(built.LIB_PATH))
    assert any("k_frame_dbk" in n for n in kds) and any("k_frame_intra" in n for n in kds)
    assert all(_tg_split(kd) == 0 for kd in kds.values())


# ---- environment switches: only the documented ones exist ----
KNOWN_ENV = {
    "H264BSDMI_LANES": "engine.hip",             # "<groups>,<heavy lanes>" of the product's flush (tests/test_gpu_api.py)
    "H264BSDMI_TAIL": "engine.hip",              # row bands / wavefronts of the per-picture kernels (tests/test_gpu_random_jobs.py via h264bsdmiDebugSetTail)
    "H264BSDMI_BAND_BUDGET": "engine.hip",
    "H264BSDMI_HEAVY_BUDGET": "engine.hip",
    "H264BSDMI_TRACE_LANES": "engine.hip",       # one diagnostic line on stderr
    "H264BSDMI_COPY_ELISION": "api.c",           # =0: every copy copied (tests/test_copy_elision.py)
    "H264BSDMI_THREADS": "api.c",                # parser pool size
    "H264BSDMI_HOST_SHARE": "api.c",             # processes that share this host's CPUs (one per GPU): the pool takes its share
    "H264BSDMI_PIN": "api.c",                    # 0 / 1 / 2: parser thread pinning
}


def test_only_documented_environment_switches_exist():
    """every getenv("H264BSDMI_*") of the product sources is in the list above and in INTEGRATION.md, and the losers of earlier
    rounds' A/B runs are not compiled in any more"""
    csrc = os.path.join(ROOT, "h264bsd_amd", "csrc")
    found = {}
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".c", ".h", ".hip")):
            for name in re.findall(r'getenv\("(H264BSDMI_\w+)"\)', open(os.path.join(csrc, f)).read()):
                found.setdefault(name, set()).add(f)
    assert set(found) == set(KNOWN_ENV), (sorted(set(found) - set(KNOWN_ENV)), sorted(set(KNOWN_ENV) - set(found)))
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for name in KNOWN_ENV:
        assert name in doc, f"{name} is not documented in INTEGRATION.md"
    engine = open(os.path.join(csrc, "engine.hip")).read()
    assert engine.count("getenv") <= 8


def test_unknown_environment_switch_is_reported(built):
    """a misspelt or retired H264BSDMI_* variable must not be silently ignored: the library names it on stderr when it is loaded"""
    code = ("import sys; sys.path.insert(0, %r); import h264bsd_amd; h264bsd_amd.lib()" % ROOT)
    env = dict(os.environ, H264BSDMI_COPY_ASIDE="1", H264BSDMI_LANES="2,1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    assert r.returncode == 0
    assert "H264BSDMI_COPY_ASIDE" in r.stderr and "unknown" in r.stderr
    assert "H264BSDMI_LANES" not in r.stderr
