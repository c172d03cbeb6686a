"""GPU: the drop-in h264bsd C API end to end (host parse -> frame jobs -> HIP engine -> host frames),
driven exactly like /root/reference/posix/test_h264bsd.c:146-177."""
import ctypes
import hashlib

import numpy as np
import pytest

from conftest import STREAMS, stream_bytes

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, scope="module")
def _through_the_product_library(built):
    """this module drives libh264bsd_mi355x.so itself — the library a C application links (exports.map) — not the harness
    library with the replay exports that the kernel tests use"""
    built.use_product_library(True)
    yield
    built.use_product_library(False)


@pytest.mark.parametrize("name", STREAMS)
def test_decode_loop_matches_reference(name, built, golden):
    g = golden[name]
    dec = built.Decoder()
    shas = []
    trace = dec.decode_stream(stream_bytes(name), on_picture=lambda f, pid, idr, err: shas.append(
        (hashlib.sha256(f.tobytes()).hexdigest(), idr, err)))
    assert [list(t) for t in trace] == g["trace"]
    assert [s[0] for s in shas] == g["frame_sha256"]
    assert [s[1] for s in shas] == [1 if i in (0, 40) else 0 for i in range(73)]
    assert all(s[2] == 0 for s in shas)
    dec.close()


def test_bgra_output_matches_reference(built, golden):
    name = "test_640x360"
    g = golden[name]
    dec = built.Decoder()
    data = stream_bytes(name)
    buf = ctypes.create_string_buffer(data, len(data))
    base, off, n = ctypes.addressof(buf), 0, 0
    while off < len(data) and n < 2:
        r, rb = dec.decode(base + off, len(data) - off)
        off += rb
        if r == built.H264BSD_PIC_RDY:
            pic = dec.next_output_picture_converted(1)
            assert hashlib.sha256(pic[0].tobytes()).hexdigest() == g["convert_sha256"][str(n)][1]
            n += 1
    dec.close()


def test_many_instances_round_robin_are_batched_and_exact(built, golden):
    """8 decoder instances advanced in lock step: every flush reconstructs 8 pictures in one tick"""
    name = "test_640x360"
    g = golden[name]
    data = stream_bytes(name)
    N = 8
    decs = [built.Decoder() for _ in range(N)]
    bufs = [ctypes.create_string_buffer(data, len(data)) for _ in range(N)]
    offs = [0] * N
    pics = [0] * N
    while any(o < len(data) for o in offs):
        ready = []
        for k, d in enumerate(decs):
            while offs[k] < len(data):
                r, rb = d.decode(ctypes.addressof(bufs[k]) + offs[k], len(data) - offs[k])
                offs[k] += rb
                assert r < built.H264BSD_ERROR
                if r == built.H264BSD_PIC_RDY:
                    ready.append(k)
                    break
        for k in ready:          # first pull flushes the whole batch
            f = decs[k].next_output_picture()
            assert hashlib.sha256(f[0].tobytes()).hexdigest() == g["frame_sha256"][pics[k]]
            pics[k] += 1
    assert pics == [73] * N
    for d in decs:
        d.close()


def _decode_n(dec, data, n_pics, pull):
    """run the decode loop, calling pull() after every PIC_RDY, until n_pics pictures came out"""
    buf = ctypes.create_string_buffer(data, len(data))
    base, off, out = ctypes.addressof(buf), 0, []
    while off < len(data) and len(out) < n_pics:
        r, rb = dec.decode(base + off, len(data) - off)
        off += rb
        if r == 1:
            while True:
                p = pull()
                if p is None:
                    break
                out.append(p)
    return out


@pytest.mark.parametrize("fmt", [3, 0, 1, 2])
@pytest.mark.parametrize("crop", [False, True])
def test_device_resident_output(built, golden, fmt, crop):
    """h264bsdmiNextOutputPictureDevice: pictures stay in HBM (zero-copy torch view), optionally cropped to the SPS
    rectangle and colour-converted on the device; contents equal the host API's (golden frames / oracle conversion)"""
    import numpy as np
    import torch
    from oracle import pyoracle
    name = "test_640x360"                     # 640x368 coded, cropped to 640x360
    g = golden[name]
    data = stream_bytes(name)
    W, H = 16 * g["width_mbs"], 16 * g["height_mbs"]
    dec = built.Decoder()
    pics = _decode_n(dec, data, 3, lambda: (lambda p: None if p is None else (p[0].clone(), p[1:]))(
        dec.next_output_picture_device(fmt, crop)))
    flag, left, cw, top, ch = dec.cropping_params()
    assert flag == 1 and (cw, ch) == (640, 360)
    dec.close()
    ref = built.Decoder()
    frames = _decode_n(ref, data, 3, ref.next_output_picture)
    ref.close()
    assert len(pics) == 3
    for (t, meta), (frame, pid, idr, err) in zip(pics, frames):
        assert t.is_cuda and t.dtype == torch.uint8 and meta == (pid, idr, err)
        x0, y0, w, h = (left, top, cw, ch) if crop else (0, 0, W, H)
        y = frame[: W * H].reshape(H, W)[y0:y0 + h, x0:x0 + w]
        c = frame[W * H:].reshape(2, H // 2, W // 2)[:, y0 // 2:(y0 + h) // 2, x0 // 2:(x0 + w) // 2]
        if fmt == 3:
            want = np.concatenate([y.reshape(-1), c.reshape(-1)])
            assert tuple(t.shape) == (h * 3 // 2, w)
            assert np.array_equal(t.cpu().numpy().reshape(-1), want)
        else:
            full = pyoracle.oracle_convert(fmt, W, H, frame).reshape(H, W)[y0:y0 + h, x0:x0 + w]
            assert tuple(t.shape) == (h, w, 4)
            assert np.array_equal(t.cpu().numpy().reshape(h, w * 4).view(np.uint32), full)


def test_parser_pool_with_async_flush_is_exact(built, golden):
    """h264bsdmiDecodePictureBatch + h264bsdmiFlushAsync: 6 instances advance picture by picture on the parser threads,
    reconstruction is enqueued without waiting; every 5th round all queued output pictures are pulled and compared"""
    names = ["test_640x360"] * 4 + ["test_1920x1080", "test_1920x1080_fullRange"]
    L = built.api_lib()
    L.h264bsdmiSetParserThreads(4)
    decs = [built.Decoder() for _ in names]
    drv = built.BatchDriver(decs, [stream_bytes(n) for n in names])
    seen = [0] * len(names)
    rounds = 0
    while True:
        ready = drv.step()
        if not ready:
            break
        assert L.h264bsdmiFlushAsync() == 0
        rounds += 1
        for k in ready:            # pictures must be pulled before the next h264bsdDecode of that instance (API contract)
            while True:
                pic = decs[k].next_output_picture()
                if pic is None:
                    break
                assert hashlib.sha256(pic[0].tobytes()).hexdigest() == golden[names[k]]["frame_sha256"][seen[k]]
                seen[k] += 1
    assert rounds == 73 and seen == [73] * len(names)
    for d in decs:
        d.close()


def test_batch_pull_on_the_parser_threads(built, golden):
    """h264bsdmiNextOutputPictureBatch: the pictures of all instances pulled at once on the library's threads (every thread waits
    for its own copies outside the engine's lock): same pictures, same ids, in every round; an instance without a picture gives NULL"""
    names = ["test_640x360"] * 5 + ["test_1920x1080", "test_1920x1080_fullRange", "test_640x360"]
    L = built.api_lib()
    L.h264bsdmiSetParserThreads(6)
    decs = [built.Decoder() for _ in names]
    drv = built.BatchDriver(decs, [stream_bytes(n) for n in names])
    seen = [0] * len(names)
    rounds = 0
    while True:
        ready = drv.step()
        if not ready:
            break
        rounds += 1
        ptrs, ids = built.pull_batch(decs)
        for k in range(len(names)):
            assert (ptrs[k] is not None and ptrs[k] != 0) == (k in ready)
            if k in ready:
                n_bytes = decs[k].pic_width() * 16 * decs[k].pic_height() * 16 * 3 // 2
                frame = np.ctypeslib.as_array(ctypes.cast(ptrs[k], ctypes.POINTER(ctypes.c_uint8)), (n_bytes,))
                assert hashlib.sha256(frame.tobytes()).hexdigest() == golden[names[k]]["frame_sha256"][seen[k]]
                assert ids[k] == seen[k]
                seen[k] += 1
        ptrs, _ = built.pull_batch(decs)                         # nothing left: NULL everywhere
        assert not any(ptrs)
    assert rounds == 73 and seen == [73] * len(names)
    for d in decs:
        d.close()


def test_lifecycle_edge_cases(built, golden):
    """things an application may do in any order: pull before anything was decoded, flush an empty decoder, shut down
    with pictures still queued and never pulled, re-init the same storage, decode again after h264bsdFlushBuffer"""
    L = built.api_lib()
    name = "test_640x360"
    g = golden[name]
    data = stream_bytes(name)
    dec = built.Decoder()
    assert dec.next_output_picture() is None            # nothing decoded yet
    dec.flush_buffer()
    assert dec.next_output_picture() is None
    assert dec.pic_width() == 0 and dec.check_valid_param_sets() == 0
    buf = ctypes.create_string_buffer(data, len(data))
    base, off, n_ready = ctypes.addressof(buf), 0, 0
    while off < len(data) and n_ready < 5:              # five pictures queued, none pulled
        r, rb = dec.decode(base + off, len(data) - off)
        off += rb
        n_ready += r == 1
    dec.close()                                         # shutdown with queued, never reconstructed pictures
    # a fresh instance on a new storage decodes the whole stream correctly afterwards
    dec = built.Decoder()
    shas = []
    dec.decode_stream(data, on_picture=lambda f, pid, idr, err: shas.append(hashlib.sha256(f.tobytes()).hexdigest()))
    assert shas == g["frame_sha256"]
    # h264bsdFlushBuffer at the end of the stream, then the same stream again on the same instance (starts with an IDR)
    dec.flush_buffer()
    assert dec.next_output_picture() is None
    shas2 = []
    dec.decode_stream(data, on_picture=lambda f, pid, idr, err: shas2.append(hashlib.sha256(f.tobytes()).hexdigest()))
    assert shas2 == g["frame_sha256"]
    dec.close()
    assert L.h264bsdmiFlush() == 0                      # nothing left anywhere


def test_concurrent_application_threads(built, golden):
    """8 application threads, each with its own decoder instances, decode and pull pictures at the same time: the engine
    serialises the device work, every picture must still be exact"""
    import threading
    name = "test_640x360"
    want = golden[name]["frame_sha256"]
    data = stream_bytes(name)
    errors = []

    def work(tid):
        try:
            for rep in range(2):
                dec = built.Decoder()
                shas = []
                dec.decode_stream(data, on_picture=lambda f, pid, idr, err: shas.append(hashlib.sha256(f.tobytes()).hexdigest()))
                dec.close()
                if shas != want:
                    errors.append((tid, rep, len(shas)))
        except Exception as e:          # noqa: BLE001
            errors.append((tid, repr(e)))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in threads), "a decoder thread is stuck"
    assert not errors, errors


def test_72_different_streams_through_the_batch_api(built, golden):
    """many DIFFERENT streams at once through the product's own scheduling — 72 decoder
    instances (60 writer streams of different picture sizes, lengths, slice structures and IDR phases, the three
    bundled streams, 9 of them started late so that their I pictures fall into other instances' P ticks), advanced
    picture by picture with h264bsdmiDecodePictureBatch on the parser threads, reconstruction enqueued with
    h264bsdmiFlushAsync.  Every output picture of every instance is compared with the reference's answer: per-frame
    SHA-256 for the bundled streams, (hash, picId, isIdr, numErrMbs) from synth_golden.json for the writer streams."""
    import json
    import os
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from h264writer import StreamWriter
    from synth_configs import CONFIGS
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "synth_golden.json")))
    names = [n for n in CONFIGS if n != "max_frame_size_4096x2304"][:60]
    streams, want, kind = [], [], []
    for n in names:
        data = StreamWriter(**CONFIGS[n]).build()
        assert hashlib.sha1(data).hexdigest() == gold[n]["stream_sha1"]
        streams.append(data); want.append([tuple(p) for p in gold[n]["pics"]]); kind.append("synth")
    for n in ["test_640x360", "test_1920x1080", "test_1920x1080_fullRange"] * 4:
        streams.append(stream_bytes(n)); want.append(golden[n]["frame_sha256"]); kind.append("bundled")
    errs_before = built.device_errors()          # (the product library's engine: no hand-built job ever ran through it, the sticky word must stay 0)
    N = len(streams)
    assert N == 72
    L = built.api_lib()
    L.h264bsdmiSetParserThreads(8)
    decs = [built.Decoder() for _ in range(N)]
    drv = built.BatchDriver(decs, streams)
    start_round = [0] * 63 + [5 * (k + 1) for k in range(9)]            # the last 9 instances join later
    got = [[] for _ in range(N)]

    def pull(k):
        while True:
            pic = decs[k].next_output_picture()
            if pic is None:
                return
            frame, pid, idr, nerr = pic
            if kind[k] == "bundled":
                got[k].append(hashlib.sha256(frame.tobytes()).hexdigest())
            else:
                got[k].append((hashlib.sha1(frame.tobytes()).hexdigest(), pid, idr, nerr))

    rounds = 0
    held = {k: (drv.size[k]) for k in range(N) if start_round[k]}     # parked: pretend they have no data yet
    for k in held:
        drv.size[k] = 0
    while True:
        for k in list(held):
            if rounds >= start_round[k]:
                drv.size[k] = held.pop(k)
        ready = drv.step()
        if not ready and not held:
            break
        assert L.h264bsdmiFlushAsync() == 0
        rounds += 1
        for k in ready:
            pull(k)
    for k in range(N):
        decs[k].flush_buffer()
        pull(k)
        assert got[k] == want[k], f"instance {k} ({kind[k]}) differs from the reference"
    assert built.device_errors() == errs_before == 0
    for d in decs:
        d.close()


@pytest.mark.gpu
@pytest.mark.parametrize("lanes", ["1,0", "2,1", "8,4"])
def test_lane_configurations_of_the_product_engine(lanes):
    """The engine's lane scheduler (stream groups + heavy lanes on their own HIP streams, per-instance ordering by
    events, engine.hip: Lane / flush_locked) in other shapes than the default 4,2: the 72-different-streams test and the
    damaged / redundant-slice fixtures (ghost and deblock-only jobs that must stay in order across lanes) in a fresh
    process with H264BSDMI_LANES set."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, H264BSDMI_LANES=lanes, GPU_MAX_HW_QUEUES="16")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                        "tests/test_gpu_api.py::test_72_different_streams_through_the_batch_api",
                        "tests/test_gpu_api.py::test_many_instances_round_robin_are_batched_and_exact",
                        "tests/test_damaged_streams.py", "-k", "not lane_configurations and (batch_api or round_robin or redundant_5 or flipped_30)"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_lane_default_follows_the_measured_stream_concurrency():
    """The default lane configuration is not guessed from the environment: the engine measures whether HIP streams run
    side by side (engine.hip: streams_run_concurrently).  With the runtime held to 4 hardware queues it must fall back to
    one lane and say so; with the library's own request for 16 it must not."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "pytest", "-q", "-x", "-s", "-m", "gpu", "-p", "no:cacheprovider",
           "tests/test_gpu_api.py::test_many_instances_round_robin_are_batched_and_exact"]
    env = {k: v for k, v in os.environ.items() if k not in ("GPU_MAX_HW_QUEUES", "H264BSDMI_LANES")}
    few = subprocess.run(cmd, cwd=root, env=dict(env, GPU_MAX_HW_QUEUES="4"), capture_output=True, text=True, timeout=600)
    assert few.returncode == 0, few.stdout[-1500:] + few.stderr[-1500:]
    assert "do not run side by side" in few.stdout + few.stderr
    many = subprocess.run(cmd, cwd=root, env=dict(env, GPU_MAX_HW_QUEUES="16"), capture_output=True, text=True, timeout=600)
    assert many.returncode == 0, many.stdout[-1500:] + many.stderr[-1500:]
    assert "do not run side by side" not in many.stdout + many.stderr


@pytest.mark.gpu
def test_too_many_lanes_are_clamped_below_the_stream_cliff():
    """More than ~12 busy HIP streams make this runtime crawl: whatever H264BSDMI_LANES asks for, the
    engine stays at 10 lanes.  A request for 16 groups + 4 heavy lanes must be clamped (and say so) and must not take more
    than twice the default configuration for the same work."""
    import os
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "pytest", "-q", "-x", "-s", "-m", "gpu", "-p", "no:cacheprovider",
           "tests/test_gpu_api.py::test_72_different_streams_through_the_batch_api"]
    env = {k: v for k, v in os.environ.items() if k != "H264BSDMI_LANES"}
    env["GPU_MAX_HW_QUEUES"] = "16"
    t0 = time.perf_counter()
    base = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    t_base = time.perf_counter() - t0
    assert base.returncode == 0, base.stdout[-1500:] + base.stderr[-1500:]
    t0 = time.perf_counter()
    many = subprocess.run(cmd, cwd=root, env=dict(env, H264BSDMI_LANES="16,4"), capture_output=True, text=True, timeout=900)
    t_many = time.perf_counter() - t0
    assert many.returncode == 0, many.stdout[-1500:] + many.stderr[-1500:]
    assert "clamped to" in many.stdout + many.stderr
    assert t_many < 2.0 * t_base + 5.0, (t_base, t_many)
