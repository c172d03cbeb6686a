#!/usr/bin/env python3
"""bench.py — throughput of the macroblock-reconstruction hot path on MI355X.

Workload (BASELINE.json configs[3] / [4]): 256 concurrent, private copies of tests/golden/test_1920x1080.h264
per GPU.  The stream is parsed ONCE on the host into packed frame jobs (capture mode), the jobs are
replicated into HBM (every stream owns its jobs and its DPB), and a "step" is one pass of the hot path
over the whole batch: 73 pictures x 256 streams = 18,688 pictures = 152.5 M macroblocks per GPU,
reconstructed (inter + intra) and deblocked by the HIP kernels.  Inputs are resident in HBM when the
timed region starts; outputs stay in HBM and are verified on the device against the reference's golden
checksums (every picture of every stream, in an untimed verification pass, and the final picture after
the timed region).  A run that is not bit-exact aborts.

`value` is the LOCK-STEP variant (every stream on the same picture index: every tick holds 256 pictures of the same
kind and cost, which is the FRIENDLIEST schedule for batched launches — the two IDR pictures of the stream give two
all-intra ticks per step, but no tick ever waits for a stray heavy picture).  The same work with odd-numbered streams
started at the second IDR ("staggered") is measured in the same run and reported under
`staggered`; streams that are not in step at all (every stream at its own picture index: what 256 independent cameras
deliver; with and without heavy lanes) under `desynchronised` — 0.32-0.65 of the lock-step figure.

One process per GPU: `python bench.py` (N=1) or
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`.  Streams are independent,
so ranks share nothing on the data path (weak scaling: 256 streams per GPU); torch.distributed (RCCL)
is used only for the barriers and the MAX-over-ranks of the elapsed time.

Output: ONE JSON line on rank 0 (see the driver contract), with `roofline` (dominant kernel, HBM bound)
and `cpu_baseline` (the compiled reference, oracle/_ref, timed on this host; N=1 only).
"""
import argparse
import json
import os
import sys
import time

# The HIP runtime maps streams onto 4 hardware queues unless told otherwise, and reads the setting when it starts;
# the stream groups / heavy lanes measured below need more (the library asks for the same when it is loaded first).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0           # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable copy)
STREAM = "test_1920x1080"


def _cpu_worker(core, seconds):
    """child process of cpu_baseline_all_cores: the compiled reference on ONE pinned core, decode-only loop with a fresh
    copy of the input per pass (it is unescaped in place); prints pictures and seconds."""
    from oracle import pyoracle
    os.sched_setaffinity(0, {core})
    data = open(os.path.join(ROOT, "tests", "golden", STREAM + ".h264"), "rb").read()
    ref = pyoracle.RefDecoder()
    ref.decode_stream(data)                                   # warm-up pass (page faults, caches)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        n += ref.decode_stream(data)[1]
    print(n, time.perf_counter() - t0, flush=True)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_quota():
    """CPUs the container may use at once (cgroup v2 cpu.max / v1 cfs quota), None when unlimited or unknown."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(p)
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / p
    except (OSError, ValueError):
        return None


def cpu_baseline_all_cores(seconds=8.0):
    """SURVEY.md §8d / posix/Rakefile:7: one pinned process per host core, all at once (the reference has no threads of
    its own; independent streams are how a host scales it)."""
    import subprocess
    cores = sorted(os.sched_getaffinity(0))
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(c), str(seconds)],
                              stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for c in cores]
    pics = fps = 0.0
    ok = 0
    for p in procs:
        out, _ = p.communicate(timeout=seconds * 6 + 120)
        try:
            n, dt = out.split()[-2:]
            pics += float(n); fps += float(n) / float(dt); ok += 1
        except (ValueError, IndexError):
            pass
    if not ok:
        return None
    quota = cpu_quota()
    return dict(value=fps * 8160, unit="macroblocks/s", fps=fps, cores=ok, cpu_quota=quota, cpu=cpu_model(),
                sample=f"{ok} pinned processes (one per hardware thread the process may run on), each looping over {STREAM}.h264 for "
                       f"{seconds:.0f} s: {int(pics)} pictures" +
                       (f"; the container's CPU quota is {quota:g} CPUs, so this is what {quota:g} CPUs of this host deliver, not {ok}" if quota and quota < ok else ""))


def cpu_baseline(data, seconds=10.0):
    """Reference decoder (oracle/_ref) on ONE host core, decode-only loop, fresh input copy per pass; plus all cores."""
    from oracle import pyoracle
    try:
        ref = pyoracle.RefDecoder()
        kind = "reference"
        run = lambda: ref.decode_stream(data)[1]
    except (FileNotFoundError, OSError):
        import h264bsd_amd
        kind = "port"

        def run():
            jobs, _, _ = h264bsd_amd.capture_stream(data)
            dpb = pyoracle.OracleDpb(jobs[0])
            for j in jobs:
                dpb.decode(j)
            return len(jobs)
    n_pics, t0, passes = 0, time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds:
        n_pics += run()
        passes += 1
    dt = time.perf_counter() - t0
    out = dict(value=n_pics * 8160 / dt, unit="macroblocks/s", fps=n_pics / dt, cores=1, kind=kind, cpu=cpu_model(),
               note="whole decoder (entropy decoding included) on the host; `value` of this bench is the pixel path only, "
                    "from frame jobs already parsed — like for like is `end_to_end` below",
               sample=f"{passes} full passes of {STREAM}.h264 ({n_pics} pictures, {dt:.1f} s) on 1 of {os.cpu_count()} host cores")
    if kind == "reference":
        out["all_cores"] = cpu_baseline_all_cores()
    return out


def end_to_end(data, streams, threads, laps=2, pull=False, barrier=None):
    """the drop-in C API end to end — host parse on the library's parser threads
    (h264bsdmiDecodePictureBatch), frame jobs built in pinned memory, one k_h2d launch per tick, kernels.  Round k+1 is parsed
    while round k reconstructs (h264bsdmiFlushAsync).  pull = False: pictures stay in HBM (what a GPU consumer gets through
    h264bsdmiNextOutputPictureDevice); pull = True: every picture is ALSO pulled to host memory with
    h264bsdNextOutputPicture semantics (layout kernel + 3.1 MB over PCIe per picture) before its instance parses on, as the
    reference's call protocol demands (posix/test_h264bsd.c:146-177) — h264bsdmiPullAndDecodePictureBatch: a parser thread pulls
    an instance's picture and then parses that instance's next one, so the link works while other instances are parsed.
    pull = "barrier": the two separate batch calls (every picture pulled, THEN the next round parsed), as measured earlier in round 5."""
    import hashlib
    import numpy as np
    import h264bsd_amd as h
    L = h.lib()
    decs = [h.Decoder() for _ in range(streams)]
    threads = L.h264bsdmiSetParserThreads(threads)
    # ("halves" / "quarters": the instances driven in 2 / 4 groups one after the other, each with its own flush, so that one group's tick runs
    # while another group's pictures cross the link — measured: 8.4 / 8.1 k fps against 8.5 k with one group; tools/experiments/hostout.py)
    parts = {"halves": 2, "quarters": 4}.get(pull, 1)
    combined = pull is True or pull in ("halves", "quarters", "whole")
    cut = [streams * p // parts for p in range(parts + 1)]
    drvs = [h.BatchDriver(decs[cut[p]:cut[p + 1]], [data * (laps + 1)] * (cut[p + 1] - cut[p])) for p in range(parts)]
    timed, t0 = 0, None
    for pic in range(73 * (laps + 1)):
        if pic == 73:                                         # first lap: untimed (pinned staging buffers are allocated)
            assert L.h264bsdmiFlush() == 0
            if barrier is not None:
                barrier()
            t0 = time.perf_counter()
        for drv in drvs:
            assert len(drv.step(pull=combined)) == drv.n
            assert not combined or pic == 0 or len(drv.pulled) == drv.n
            assert L.h264bsdmiFlushAsync() == 0
        if pull == "barrier":
            ptrs, _ = h.pull_batch(decs)
            assert all(ptrs)
        timed += pic >= 73
    jobs, _, info = h.capture_stream(data, copy_elision=os.environ.get("H264BSDMI_COPY_ELISION", "1")[:1] != "0")
    frame_bytes = info["width_mbs"] * info["height_mbs"] * 384
    last = None
    if combined:
        last, _ = h.pull_batch(decs, frame_bytes)             # the last round's pictures
        assert all(v is not None for v in last)
    assert L.h264bsdmiFlush() == 0
    dt = time.perf_counter() - t0
    # Verification (outside the timed region): the LAST picture of every instance — pulled to host memory through
    # the same batch call — must be the reference's picture 72 (sha256 of the first instance's, byte-for-byte equality of all the
    # others with it).  Every picture of the final group of pictures feeds it through inter prediction, and the kernel-only legs
    # check every single picture of every stream on the device.
    if last is None:
        last, _ = h.pull_batch(decs, frame_bytes)
        assert all(v is not None for v in last)
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "golden.json")))[STREAM]["frame_sha256"][-1]
    if hashlib.sha256(last[0].tobytes()).hexdigest() != want:
        raise SystemExit("end_to_end: the last picture of instance 0 is not the reference's (sha256)")
    bad = [i for i in range(1, streams) if not np.array_equal(last[i], last[0])]
    if bad:
        raise SystemExit(f"end_to_end: the last picture of {len(bad)} instances differs from the reference's (first: instance {bad[0]})")
    h2d = sum(len(j) for j in jobs) * streams * laps
    for d in decs:
        d.close()
    pics = streams * timed
    return dict(value=pics * 8160 / dt, unit="macroblocks/s", fps=pics / dt, streams=streams, parser_threads=int(threads),
                host_cores=os.cpu_count(), cpu_quota=cpu_quota(), h2d_bytes_per_picture=h2d / pics,
                d2h_bytes_per_picture=info["width_mbs"] * info["height_mbs"] * 384 if pull else 0, seconds=dt,
                device_errors=h.device_errors(), verified=f"last picture of all {streams} instances sha256 / byte-equal to the reference's picture 72",
                sample=f"{streams} decoder instances x {timed} pictures through h264bsdDecode-equivalent batch calls, PCIe inclusive, "
                       f"{dt:.1f} s; " + ("every picture pulled to host memory (h264bsdNextOutputPicture semantics; h264bsdmiPullAndDecodePictureBatch, which runs on min(parser_threads, CPUs the process may have running at once) threads)" if pull else "pictures left in HBM"))


def main():
    if len(sys.argv) == 4 and sys.argv[1] == "--cpu-worker":
        return _cpu_worker(int(sys.argv[2]), float(sys.argv[3]))
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=80, help="timed steps of the lock-step variant (default: ~10 s of steady state, SURVEY.md 8d config 4)")
    ap.add_argument("--side-steps", type=int, default=0, help="timed steps of the other variants (staggered, ARGB, desynchronised); 0 = min(--steps, 20)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--streams", type=int, default=256, help="concurrent streams per GPU")
    ap.add_argument("--groups", type=int, default=1, help="stream groups on separate HIP streams (overlap)")
    ap.add_argument("--ramp-seconds", type=float, default=4.0, help="untimed load before the warm-up steps (device clock ramp)")
    ap.add_argument("--time-all-kernels", action="store_true", help="bracket all five kernels with events in the timed steps too (A/B of the event overhead)")
    ap.add_argument("--no-copy-elision", action="store_true", help="capture the replayed frame jobs without copy elision (the product's default with a device is ON: copies that would rewrite what the destination frame holds are left out, include/h264bsd_mi355x.h)")
    ap.add_argument("--no-full-copies-variant", action="store_true", help="skip the lock-step measurement without copy elision (reported next to `value`)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-groups-variant", action="store_true", help="skip the lock-step measurement with 4 stream groups")
    ap.add_argument("--no-desync", action="store_true", help="skip the fully desynchronised variants (reported next to the lock-step value)")
    ap.add_argument("--no-staggered", action="store_true", help="skip the staggered-start variant (reported next to the lock-step value)")
    ap.add_argument("--no-argb", action="store_true", help="skip the config-3 variant (colour conversion of every picture inside the timed region)")
    ap.add_argument("--argb-no-hosting", action="store_true", help="config-3 variant with a k_convert_tiles launch behind every tick instead of the conversion hosted by the next tick's k_frame_dbk (A/B)")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the end-to-end leg through the drop-in C API (host parse + H2D + kernels)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: start the N ranks here (one process per GPU), exactly as the driver's
        # `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` would; rank 0 prints the line
        import socket
        import subprocess
        rc = 1
        for attempt in range(2):                      # (a second rendezvous port if the first try dies: ports are picked, not reserved)
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
                   "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
            rc = subprocess.call(cmd)
            if rc == 0:
                break
        raise SystemExit(rc)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but {world} rank(s) were started (WORLD_SIZE): refusing to report a line for another GPU count")
    side_steps = args.side_steps or min(args.steps, 20)
    # ranks on this host share its CPUs: the library's parser pool takes its share
    os.environ.setdefault("H264BSDMI_HOST_SHARE", os.environ.get("LOCAL_WORLD_SIZE", str(world)))

    import torch
    import h264bsd_amd

    if not torch.cuda.is_available() or h264bsd_amd.device_count() <= 0:
        raise SystemExit("bench.py needs an MI355X: the product has no CPU pixel path")
    # one process per GPU.  When there are fewer GPUs than ranks (the 2-ranks-on-one-GPU smoke test of the N > 1 path)
    # the ranks share devices and the three scalar reductions go over gloo: RCCL cannot put two ranks on one device.
    n_dev = torch.cuda.device_count()
    device = local_rank % n_dev
    torch.cuda.set_device(device)
    assert h264bsd_amd.lib().h264bsdmiSetDevice(device) == 0
    dist = None
    red_dev = "cuda"
    if world > 1:
        import torch.distributed as dist
        if n_dev >= world:
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group("gloo")
            red_dev = "cpu"

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    gdir = os.path.join(ROOT, "tests", "golden")
    golden = json.load(open(os.path.join(gdir, "golden.json")))[STREAM]
    data = open(os.path.join(gdir, STREAM + ".h264"), "rb").read()
    elide = not args.no_copy_elision
    jobs, _, info = h264bsd_amd.capture_stream(data, copy_elision=elide)          # host parse, once
    heads = [h264bsd_amd.job_header(j) for j in jobs]
    copy_mbs = sum(h["n_copy_mbs"] for h in heads)
    copy_mbs_full = sum(h264bsd_amd.job_header(j)["n_copy_mbs"] for j in h264bsd_amd.capture_stream(data)[0]) if elide else copy_mbs
    n_pics, n_mbs = len(jobs), heads[0]["n_mbs"]
    kernels = h264bsd_amd.Replay.KERNELS
    BESIDE = ("k_copy", "k_dbk")                 # launched on side streams next to k_recon_inter (engine.hip launch_tick)
    golden_sums = golden["frame_checksum64"]

    # ONE replay set (jobs + DPBs resident in HBM: 22 GB for 256 x 1080p) serves every leg that replays these jobs — lock-step,
    # staggered, + conversion, the three desynchronised schedules: a leg asks for its schedule (Replay.reschedule: new
    # descriptors, zeroed frame buffers), nothing is allocated or uploaded twice.  Only the leg without copy elision replays
    # OTHER jobs and builds its own set.
    shared = {"rep": None}

    def replay_for(these_jobs, **sched):
        if these_jobs is jobs:
            if shared["rep"] is None:
                shared["rep"] = h264bsd_amd.Replay(jobs, n_streams=args.streams, **sched)
            else:
                shared["rep"].reschedule(**sched)
            return shared["rep"]
        return h264bsd_amd.Replay(these_jobs, n_streams=args.streams, **sched)

    def release(rep):
        if rep is not shared["rep"]:
            rep.close()

    def run_variant(odd_offset, steps, jobs=jobs, main=True):
        """Verify, warm up and time one variant of the workload.  odd_offset = 0: lock-step (every stream on
        the same picture index: two all-IDR ticks per step, every tick homogeneous); otherwise odd streams start at the
        second IDR (SURVEY.md §8d config 4 "staggered").  Returns (elapsed s, kernel ms, launches, device ms, job bytes)."""
        rep = replay_for(jobs, odd_offset=odd_offset)
        extra = {}

        def verify(i):
            # picture i of the even streams and picture (i + odd_offset) % n of the odd streams
            for parity, p in ((0, i), (1, (i + odd_offset) % n_pics)):
                if parity and (not odd_offset or args.streams < 2):
                    continue
                sums = rep.checksums(heads[p]["cur_slot"])
                sel = sums if not odd_offset else sums[parity::2]
                if not (sel == golden_sums[p]).all():
                    raise SystemExit(f"rank {rank}: picture {p} is not bit-exact on {(sel != golden_sums[p]).sum()} streams")

        # ---- untimed verification pass: every picture of every stream against the reference ----
        for i in range(n_pics):
            rep.run(i, 1)
            verify(i)
        rep.set_groups(args.groups)
        # untimed clock ramp: a fresh box reaches its steady clocks only after a few seconds of load (the first
        # bench run on a new box measured 4-5 % low with one warm-up step); W warm-up steps follow as asked.
        # These untimed passes bracket ALL five kernels with HIP events: the per-kernel breakdown of a step and
        # the choice of the dominant kernel come from the last of them.
        breakdown = None
        t_ramp = time.perf_counter()
        while time.perf_counter() - t_ramp < (args.ramp_seconds if (main and odd_offset == 0) else min(args.ramp_seconds, 1.0)):   # (the side variants start on a warm GPU)
            rep.run()
            breakdown = rep.timings()
        for _ in range(args.warmup):
            rep.run()
            breakdown = rep.timings()
        # timed steps: events only around the dominant kernel (every event is a barrier packet between two kernels)
        # (k_copy and k_dbk run on HIP streams of their own beside k_recon_inter: their event-to-event times are stretched by what they share
        # the device with and are not part of the step's critical path — the dominant kernel is chosen among the kernels of the tick's own stream)
        dom = max((k for k in kernels if k not in BESIDE), key=lambda k: breakdown[k][0]) if breakdown else None
        if dom is not None and not args.time_all_kernels:
            rep.set_timed_kernels(1 << kernels.index(dom))
        barrier()
        t0 = time.perf_counter()
        k_ms = {k: 0.0 for k in kernels}
        k_n = {k: 0 for k in kernels}
        dev_total_ms = 0.0
        for _ in range(steps):
            rep.run()
            t = rep.timings()        # waits for the step; HIP events recorded on the engine's own stream
            for k in kernels:
                if t[k][0] is not None:
                    k_ms[k] += t[k][0]
                k_n[k] += t[k][1]
            dev_total_ms += t["total_ms"]
        barrier()
        elapsed = time.perf_counter() - t0
        # the same lock-step work with the streams split into 4 groups that run their ticks on their own HIP streams
        # (the per-picture kernels of one group overlap with the other groups' work); reported next to `value`
        if main and odd_offset == 0 and args.groups == 1 and args.streams >= 8 and not args.no_groups_variant:
            rep.set_groups(4)
            rep.run(); rep.sync()
            barrier()
            tg = time.perf_counter()
            for _ in range(side_steps):
                rep.run()
                rep.timings()
            barrier()
            extra["groups4_elapsed"] = time.perf_counter() - tg
            rep.set_groups(1)
        rep.set_timed_kernels(31)
        if dom is None:
            dom = max((k for k in kernels if k not in BESIDE), key=lambda k: k_ms[k])
            breakdown = {k: (k_ms[k] / steps, k_n[k] // steps) for k in kernels}
        verify(n_pics - 1)                                     # the final pictures, after the timed region
        job_bytes = rep.job_bytes
        release(rep)
        local = elapsed
        if dist is not None:
            tt = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        return elapsed, k_ms, k_n, dev_total_ms, job_bytes, dom, breakdown, local, extra

    elapsed, k_ms, k_n, dev_total_ms, job_bytes, dom, breakdown, local_elapsed, lock_extra = run_variant(0, args.steps)
    # the same lock-step work with EVERY copy macroblock copied (no copy elision): what the number is without that
    # optimisation of the parser, on the same line
    full_copies = None
    if elide and not args.no_full_copies_variant:
        fc_elapsed, _, _, fc_dev_ms, _, _, fc_breakdown, _, _ = run_variant(0, side_steps, jobs=h264bsd_amd.capture_stream(data)[0], main=False)
        full_copies = dict(elapsed=fc_elapsed, breakdown=fc_breakdown, dev_ms=fc_dev_ms)
    staggered = None
    if not args.no_staggered and args.streams > 1:
        idr = [i for i, h in enumerate(heads) if h["is_idr"] and i > 0]
        if idr:
            st_elapsed, _, _, st_dev_ms, _, _, st_breakdown, _, _ = run_variant(idr[0], side_steps)
            staggered = dict(odd_stream_offset=idr[0], elapsed=st_elapsed, breakdown=st_breakdown, dev_ms=st_dev_ms)

    # ---- BASELINE.json config 3: the same lock-step work with the colour conversion of every produced picture inside
    # the timed region (BGRA = the reference's "ARGB" word, h264bsdConvertToBGRA semantics): +1024 B written per macroblock.
    # The pictures of tick i - 1 are converted by wavefronts of tick i's k_frame_dbk workgroups, beside the filtering of
    # picture i (kernels/convert.hip.h); the last picture of a step, and a picture whose successor is decoded into the
    # same frame buffer, by a k_convert_tiles launch.  Checked against the reference's own conversion (golden.json
    # convert_sha256) in untimed passes first: pictures 0, 1 and 72 through the launch, pictures 0 and 1 as the hosts of
    # ticks 1 and 2 leave them in the conversion buffer.
    argb = None
    if not args.no_argb:
        import hashlib
        rep = replay_for(jobs)
        w_px, h_px = info["width_mbs"] * 16, info["height_mbs"] * 16
        for i in range(n_pics):
            rep.run(i, 1)
            if str(i) in golden["convert_sha256"]:
                rep.convert(heads[i]["cur_slot"], h264bsd_amd.FMT_BGRA)
                got = rep.fetch_converted(args.streams - 1, w_px * h_px)
                if hashlib.sha256(got.tobytes()).hexdigest() != golden["convert_sha256"][str(i)][h264bsd_amd.FMT_BGRA]:
                    raise SystemExit(f"rank {rank}: BGRA conversion of picture {i} differs from the reference")
        hosted_checked = []
        for i in sorted(int(k) for k in golden["convert_sha256"]):
            if i + 1 >= n_pics or heads[i + 1]["cur_slot"] == heads[i]["cur_slot"]:
                continue
            rep.set_convert(h264bsd_amd.FMT_BGRA, trailing=False)     # ticks 0 .. i + 1, no launch behind the last: the buffer holds picture i as tick i + 1's hosts wrote it
            rep.run(0, i + 2); rep.sync()
            for s_ in (0, args.streams - 1):
                got = rep.fetch_converted(s_, w_px * h_px)
                if hashlib.sha256(got.tobytes()).hexdigest() != golden["convert_sha256"][str(i)][h264bsd_amd.FMT_BGRA]:
                    raise SystemExit(f"rank {rank}: BGRA conversion of picture {i} by the hosts of tick {i + 1} differs from the reference (stream {s_})")
            hosted_checked.append(i)
        rep.set_convert(h264bsd_amd.FMT_BGRA, hosting=not args.argb_no_hosting)
        rep.run(); rep.sync()
        barrier()
        t0 = time.perf_counter()
        conv_ms, conv_n, a_dev_ms = 0.0, 0, 0.0
        for _ in range(side_steps):
            rep.run()
            a_dev_ms += rep.timings()["total_ms"]
            ms, n = rep.convert_timings()
            conv_ms += ms; conv_n += n
        barrier()
        a_elapsed = time.perf_counter() - t0
        sums = rep.checksums(heads[-1]["cur_slot"])
        if not (sums == golden_sums[-1]).all():
            raise SystemExit(f"rank {rank}: ARGB variant: final pictures are not bit-exact")
        # ... and what the timed region converted last (the launch behind the step's last tick) is the reference's conversion
        if str(n_pics - 1) in golden["convert_sha256"]:
            got = rep.fetch_converted(0, w_px * h_px)
            if hashlib.sha256(got.tobytes()).hexdigest() != golden["convert_sha256"][str(n_pics - 1)][h264bsd_amd.FMT_BGRA]:
                raise SystemExit(f"rank {rank}: ARGB variant: the conversion inside the timed region differs from the reference")
        rep.set_convert(-1)
        release(rep)
        if dist is not None:
            tt = torch.tensor([a_elapsed], dtype=torch.float64, device=red_dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            a_elapsed = float(tt.item())
        argb = dict(elapsed=a_elapsed, conv_ms=conv_ms, conv_n=conv_n, dev_ms=a_dev_ms, hosted_checked=hosted_checked)

    # ---- streams that are not in step at all: stream s starts at picture s * n_pics / n_streams.  Every tick then
    # holds I pictures AND the heaviest P pictures of the stream, and a tick lasts as long as its slowest picture:
    # (a) common ticks, (b) mostly-intra pictures on 4 extra HIP streams ("heavy lanes"), rejoining 4 ticks later,
    # (c) 3 heavy lanes and the streams split into 9 groups that run their own ticks on their own HIP streams (12 busy
    # HIP streams: more fall off a cliff on this runtime, tools/probes/queue_probe.hip).
    # Verified at the end of laps 2 and N (a lap = every stream n_pics pictures; the last picture of stream s is
    # picture offsets[s] - 1, so the 256 streams together cover every picture index).
    desync = None
    if not args.no_desync and args.streams > 1:
        offsets = [(st * n_pics) // args.streams for st in range(args.streams)]
        slots = sorted(set(h["cur_slot"] for h in heads))
        desync = {"offsets": "stream s starts at picture floor(s * n_pics / n_streams)"}
        def desync_leg(key, lanes, delay, groups, cliff_ms=None):
            """one schedule of the desynchronised set; None when a lap takes longer than cliff_ms (the runtime's stream cliff, below)"""
            rep = replay_for(jobs, offsets=offsets, heavy_lanes=lanes, heavy_delay=delay, groups=groups)

            def verify_lap():
                sums = {sl: rep.checksums(sl) for sl in slots}
                for st in range(args.streams):
                    last = (offsets[st] - 1) % n_pics
                    if int(sums[heads[last]["cur_slot"]][st]) != golden_sums[last]:
                        raise SystemExit(f"rank {rank}: desynchronised set ({key}): stream {st} is not bit-exact")
            t0 = time.perf_counter()
            rep.run(); rep.sync()
            lap_ms = (time.perf_counter() - t0) * 1e3               # (the first lap: a schedule on the cliff is given up after one of them)
            if cliff_ms is not None:
                slow = lap_ms > max(cliff_ms, 1500.0)      # (a first lap also pays for new HIP streams: cliff laps measured 2-21 s, good ones 0.13-0.4 s)
                if dist is not None:                      # (every rank takes the same branch)
                    tt = torch.tensor([1.0 if slow else 0.0], dtype=torch.float64, device=red_dev)
                    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                    slow = tt.item() > 0
                if slow:
                    release(rep)
                    return {"lap_ms": lap_ms}
            rep.run(); rep.sync()
            verify_lap()
            barrier()
            t0 = time.perf_counter()
            for _ in range(side_steps):
                rep.run()
            rep.sync()
            barrier()
            dt = time.perf_counter() - t0
            verify_lap()
            release(rep)
            if dist is not None:
                tt = torch.tensor([dt], dtype=torch.float64, device=red_dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                dt = float(tt.item())
            return {"value": n_pics * args.streams * world * n_mbs * side_steps / dt, "unit": "macroblocks/s",
                    "ms_per_step": dt * 1e3 / side_steps, "steps": side_steps, "lanes": lanes, "rejoin_after_ticks": delay, "stream_groups": groups}

        desync["common_ticks"] = desync_leg("common_ticks", 0, 0, 1)
        desync["heavy_lanes"] = desync_leg("heavy_lanes", 4, 4, 1)
        # Stream groups on lanes of their own next to the heavy lanes: the more groups, the better the groups' ticks fit their pictures — until the
        # HIP runtime falls off its stream cliff (above ~12 busy HIP streams a lap takes seconds instead of ~0.15 s, docs/EXPERIMENTS.md; where the
        # edge lies depends on which hardware queues the process's streams landed on).  A lap slower than four laps of the plain heavy-lane
        # schedule is taken for the cliff: the leg steps down to fewer groups and says so.
        cliff_ms = 4.0 * desync["heavy_lanes"]["ms_per_step"]
        fell = []
        for groups in (9, 6, 4):
            leg = desync_leg("heavy_lanes_stream_groups", 3, 4, groups, cliff_ms=cliff_ms)
            if "value" in leg:
                if fell:
                    leg["stream_cliff"] = {"groups_tried_first": fell, "note": "those schedules took seconds per lap on this process's HIP streams (runtime stream cliff): not timed"}
                desync["heavy_lanes_stream_groups"] = leg
                break
            fell.append({"stream_groups": groups, "lap_ms": leg["lap_ms"]})
        if "heavy_lanes_stream_groups" not in desync:
            desync["stream_cliff"] = fell                       # (a list, not a leg: every schedule with stream groups took seconds per lap)

    if shared["rep"] is not None:
        shared["rep"].close()
        shared["rep"] = None

    # on-box ceiling of a plain device-to-device copy (report the fraction of both peaks)
    copy_gbs = None
    if rank == 0:
        src = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
        dst = torch.empty_like(src)
        dst.copy_(src)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            dst.copy_(src)
        e1.record()
        torch.cuda.synchronize()
        copy_gbs = 10 * 2 * src.numel() / (e0.elapsed_time(e1) * 1e-3) / 1e9      # read + write
        del src, dst

    # per-GPU values (config 5 asks for them next to the node total): every rank's own elapsed time of the timed steps
    per_gpu = None
    if dist is not None:
        mine = torch.tensor([local_elapsed], dtype=torch.float64, device=red_dev)
        allv = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allv, mine)
        per_gpu = [n_pics * args.streams * n_mbs * args.steps / float(v.item()) for v in allv]

    # ---- end to end through the drop-in C API, on EVERY rank at once ("end-to-end MB/s through the C API at 1/2/4/8
    # GPUs"): each rank's library sizes its parser pool for its share of the CPUs the container may use (H264BSDMI_HOST_SHARE =
    # ranks on this host) and pins it to its GPU's NUMA node; the node figure is all ranks' pictures over the slowest rank's time.
    e2e = {}
    if not args.no_end_to_end:
        for key, pull, laps in (("end_to_end", False, 2), ("end_to_end_host_output", True, 1)):
            barrier()
            leg = end_to_end(data, min(args.streams, 256), 0, laps=laps, pull=pull, barrier=barrier if dist is not None else None)      # 0 threads: the library's default
            if dist is not None:
                mine = torch.tensor([leg["fps"], leg["seconds"], float(leg["parser_threads"])], dtype=torch.float64, device=red_dev)
                allv = [torch.zeros_like(mine) for _ in range(world)]
                dist.all_gather(allv, mine)
                secs = max(float(v[1].item()) for v in allv)
                pics = sum(float(v[0].item()) * float(v[1].item()) for v in allv)
                leg["per_gpu"] = {"fps": [float(v[0].item()) for v in allv], "parser_threads": [int(v[2].item()) for v in allv],
                                  "note": "each rank's own clock; the ranks share the host's CPUs (and its CPU quota, if any)"}
                leg["fps"] = pics / secs
                leg["value"] = pics * 8160 / secs
                leg["n_gpus"] = world
                leg["streams"] = leg["streams"] * world
            e2e[key] = leg

    if rank == 0:
        pics_per_step = n_pics * args.streams * world
        mbs = pics_per_step * n_mbs * args.steps
        # algorithmic bytes: 384 B written per MB + 384 B of reference read per inter MB +
        # the packed syntax actually consumed (the frame jobs)
        n_inter = sum(h["n_inter"] for h in heads)
        alg_bytes_stream = 384 * n_mbs * n_pics + 384 * n_inter + job_bytes
        alg_per_mb = alg_bytes_stream / (n_mbs * n_pics)
        # dominant kernel: the one with the largest share of device time (chosen in the warm-up passes, where all five
        # kernels are bracketed by events; in the timed region only this one is)
        launches = max(k_n[dom], 1)
        avg_launch_us = k_ms[dom] * 1e3 / launches
        # every launch of every kernel covers one picture of every stream: the units of one launch are the
        # macroblocks of one tick (SURVEY.md §8d per-MB figure x MBs per launch)
        units_per_launch = n_mbs * n_pics * args.streams * args.steps / launches
        achieved = alg_per_mb * units_per_launch / (avg_launch_us * 1e-6) / 1e9       # GB/s
        path_gbs = alg_bytes_stream * args.streams * args.steps / (dev_total_ms * 1e-3) / 1e9
        # `frac` above charges the WHOLE path's algorithmic bytes to the dominant kernel's time (the contract's formula).  frac_own prices
        # every kernel against the bytes ITS OWN work has to move: tiles read and written by the macroblocks on
        # its list, the records and coefficient blocks it consumes — per stream pass, from the frame-job headers.
        n_coef_blocks = sum(h["n_coef_blocks"] for h in heads)
        gen_mbs, intra_mbs, dbk_mbs = sum(h["n_gen"] for h in heads), sum(h["n_intra"] for h in heads), sum(h["n_dbk"] for h in heads)
        coded_share = 32.0 * n_coef_blocks / max(1, gen_mbs + intra_mbs)         # coefficient bytes per reconstructed (non-copy) macroblock
        own_bytes = {"k_copy": 768.0 * copy_mbs,
                     "k_recon_inter": gen_mbs * (768.0 + 32.0 + coded_share),       # reference window + tile out + record + coefficients
                     "k_dbk": dbk_mbs * (3 * 32.0 + 48.0 + 1.0),                     # own and two neighbour records in, deblocking record + flag out
                     "k_frame_intra": intra_mbs * (384.0 + 32.0 + coded_share),
                     "k_frame_dbk": 2 * dbk_mbs * 48.0 + dbk_mbs * 768.0}           # tile in and out (luma and chroma graph), the record for each graph
        frac_own = {k: (own_bytes[k] * args.streams / (breakdown[k][0] * 1e-3) / 1e9 / HBM_PEAK_GBS if breakdown[k][0] > 0 else None) for k in kernels}
        # HBM bytes per launch of the dominant kernel: PMC counters need rocprofv3 (separate --pmc passes,
        # tools/refresh_profiles.sh), so they cannot be collected inside this run; the committed table is only used
        # while it belongs to the kernels that just ran (sha256 of the kernel sources), otherwise traffic is null
        traffic, traffic_note, traffic_all, traffic_whole = None, "no traffic table in profiles/", None, None
        try:
            import glob
            tpath = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))[-1]
            tname = os.path.relpath(tpath, ROOT)
            tfile = json.load(open(tpath))
            from h264bsd_amd.srchash import kernel_source_sha256
            src_sha = kernel_source_sha256(ROOT)
            if tfile.get("kernel_source_sha256") != src_sha:
                traffic_note = f"{tname} was measured with other kernel sources: stale, not reported (rerun tools/refresh_profiles.sh)"
            else:
                per_kernel = {k: v.get("fetch_bytes_per_launch_calibrated", v["fetch_bytes_per_launch"]) +
                              v.get("write_bytes_per_launch_calibrated", v["write_bytes_per_launch"]) for k, v in tfile["kernels"].items() if k in kernels}
                traffic = per_kernel[dom]
                traffic_all = per_kernel
                traffic_whole = sum(per_kernel.values())          # every kernel is launched once per tick: HBM bytes of one tick of the whole path
                traffic_note = f"{tname}: FETCH_SIZE + WRITE_SIZE per launch, separate rocprofv3 --pmc passes of this command, calibrated on k_copy's known byte count"
        except (OSError, KeyError, ValueError, IndexError):
            pass
        # the other roofline: wave-level vector instructions.  Counts per launch come from separate rocprofv3 --pmc passes
        # (tools/sq_profile.sh -> profiles/rNN_sq_counters.json, valid for these kernel sources only), the cost of an instruction
        # from tools/probes/valu_rate_probe.hip, the launch durations from THIS run (warm-up pass, all kernels bracketed)
        valu = None
        try:
            import glob
            vpath = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_sq_counters.json")))[-1]
            vfile = json.load(open(vpath))
            from h264bsd_amd.srchash import kernel_source_sha256
            if vfile.get("kernel_source_sha256") != kernel_source_sha256(ROOT):
                valu = {"note": f"{os.path.relpath(vpath, ROOT)} was measured with other kernel sources: stale, not reported (rerun tools/sq_profile.sh)"}
            else:
                cyc = float(vfile["cycles_per_wave_instruction"])
                peak = 1024 * 2.4e9 / cyc                                     # 256 CUs x 4 SIMDs, wave64 instructions per second
                scale = args.streams / float(vfile.get("streams", 256))
                per = {}
                for k in kernels:
                    if k in vfile["kernels"] and breakdown[k][1]:
                        n_i = vfile["kernels"][k]["valu_wave_instr_per_launch"] * scale
                        t_s = breakdown[k][0] * 1e-3 / breakdown[k][1]
                        per[k] = {"wave_instr_per_launch": n_i, "avg_launch_us": t_s * 1e6, "achieved": n_i / t_s, "frac": n_i / t_s / peak,
                                  "wave_instr_per_macroblock": n_i / (n_mbs * args.streams),
                                  "lane_util": (vfile["kernels"][k].get("active_lanes_per_valu_instr") or 0.0) / 64.0}
                tot_i = sum(v["wave_instr_per_launch"] for v in per.values())
                valu = {"peak_wave_instr_per_s": peak, "cycles_per_wave_instruction": cyc, "source": os.path.relpath(vpath, ROOT),
                        "per_kernel": per, "whole_path": {"wave_instr_per_tick": tot_i, "achieved": tot_i * n_pics * args.steps / (dev_total_ms * 1e-3),
                                                          "frac": tot_i * n_pics * args.steps / (dev_total_ms * 1e-3) / peak,
                                                          "note": "every kernel runs once per tick; k_copy and k_dbk run on streams of their own beside k_recon_inter, so the kernel times add up to more than the total"}}
        except (OSError, KeyError, ValueError, IndexError, ZeroDivisionError):
            pass
        moved_per_mb = alg_per_mb - 768.0 * (copy_mbs_full - copy_mbs) / (n_mbs * n_pics)
        out = {
            "metric": "1080p macroblocks/s", "value": mbs / elapsed, "unit": "macroblocks/s",
            "fps": pics_per_step * args.steps / elapsed,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{args.streams} concurrent copies of {STREAM}.h264 per GPU "
                                   f"({n_pics} pictures x {n_mbs} MB), kernel-only replay from HBM-resident frame jobs, "
                                   "inter+intra reconstruction + in-loop deblocking, bit-exact vs reference verified on device",
                       "streams_per_gpu": args.streams, "pictures_per_step": pics_per_step, "parallelism": f"streams/{world}",
                       "stream_groups": args.groups,
                       "row_bands": os.environ.get("H264BSDMI_TAIL", "library default (engine.hip TailConfig)"),
                       "copy_elision": {"on": elide, "copy_mbs_per_stream_pass": copy_mbs_full, "elided": copy_mbs_full - copy_mbs,
                                        "note": "the parser leaves out whole-tile copies whose destination frame buffer already holds the "
                                                "source's bytes (unchanged since the buffer's previous picture, untouched by deblocking): "
                                                "same pictures, verified on device; alg_bytes_per_mb below still counts them (SURVEY 8d formula), "
                                                "alg_bytes_per_mb_moved does not"}},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "frac_note": "`frac` = the contract's formula: the WHOLE path's algorithmic bytes per launch / the dominant kernel's launch time / peak; "
                                      "`frac_own` = the dominant kernel's OWN algorithmic bytes / its own time / peak; `whole_path_frac` = the whole path's bytes / the whole step / peak",
                         "frac_own": frac_own[dom], "frac_own_per_kernel": frac_own,
                         "own_alg_bytes_per_stream_pass": own_bytes,
                         "traffic": traffic, "traffic_source": traffic_note,
                         "traffic_per_kernel": traffic_all, "traffic_whole_path": traffic_whole,
                         "traffic_ratio": (traffic_whole / (alg_per_mb * units_per_launch)) if traffic_whole else None,
                         "alg_bytes_per_launch": alg_per_mb * units_per_launch,
                         "alg_bytes_per_mb": alg_per_mb, "alg_bytes_per_mb_moved": alg_per_mb - 768.0 * (copy_mbs_full - copy_mbs) / (n_mbs * n_pics),
                         "mbs_per_launch": units_per_launch,
                         "avg_launch_us": avg_launch_us, "launches": launches,
                         "whole_path_GBs": path_gbs, "whole_path_frac": path_gbs / HBM_PEAK_GBS,
                         "whole_path_GBs_moved": path_gbs * moved_per_mb / alg_per_mb, "whole_path_frac_moved": path_gbs * moved_per_mb / alg_per_mb / HBM_PEAK_GBS,
                         "valu": valu,
                         "copy_ceiling_GBs": copy_gbs, "frac_of_copy_ceiling": achieved / copy_gbs,
                         "whole_path_frac_of_copy_ceiling": path_gbs / copy_gbs,
                         "device_ms_per_step": dict({k: breakdown[k][0] for k in kernels}, total=dev_total_ms / args.steps,
                                                    note="per-kernel: last untimed warm-up pass (all kernels bracketed by events); total: timed steps; k_copy and k_dbk run on streams of their own beside k_recon_inter (its time includes their interference), so the kernel times add up to more than the total"),
                         "launches_per_step": {k: k_n[k] // args.steps for k in kernels}},
        }
        if "groups4_elapsed" in lock_extra:
            ge = lock_extra["groups4_elapsed"]
            out["lock_step_4_stream_groups"] = {"value": n_pics * args.streams * n_mbs * side_steps / ge, "unit": "macroblocks/s",
                                                "ms_per_step": ge * 1e3 / side_steps, "steps": side_steps,
                                                "note": "rank 0's own clock; same work as `value`, streams split into 4 groups on 4 HIP streams"}
        if full_copies is not None:
            out["lock_step_without_copy_elision"] = {
                "value": pics_per_step * n_mbs * side_steps / full_copies["elapsed"], "unit": "macroblocks/s",
                "ms_per_step": full_copies["elapsed"] * 1e3 / side_steps, "steps": side_steps,
                "device_ms_per_step": dict({k: full_copies["breakdown"][k][0] for k in kernels}, total=full_copies["dev_ms"] / side_steps),
                "note": "frame jobs captured with h264bsdmiSetCopyElision(0): every copy macroblock is copied; verified on device like `value`"}
        if staggered is not None:
            # same work per step, odd streams start at the second IDR: I pictures never fill a whole tick
            side_mbs = pics_per_step * n_mbs * side_steps
            st_gbs = alg_bytes_stream * args.streams * world * side_steps / staggered["elapsed"] / 1e9
            out["staggered"] = {"value": side_mbs / staggered["elapsed"], "unit": "macroblocks/s",
                                "fps": pics_per_step * side_steps / staggered["elapsed"],
                                "ms_per_step": staggered["elapsed"] * 1e3 / side_steps, "steps": side_steps,
                                "odd_stream_offset_pictures": staggered["odd_stream_offset"],
                                "vs_lock_step": side_mbs / staggered["elapsed"] / (mbs / elapsed),
                                "whole_path_GBs": st_gbs, "whole_path_frac": st_gbs / HBM_PEAK_GBS / world,
                                "whole_path_frac_moved": st_gbs * moved_per_mb / alg_per_mb / HBM_PEAK_GBS / world,
                                "device_ms_per_step": dict({k: staggered["breakdown"][k][0] for k in kernels},
                                                           total=staggered["dev_ms"] / side_steps)}
        if argb is not None:
            # config 3: algorithmic bytes per macroblock + 1024 (32-bit pixels written); k_convert's own roofline:
            # 384 B read + 1024 B written per macroblock, every launch covers one picture of every stream
            a_launches = max(argb["conv_n"], 1)
            conv_us = argb["conv_ms"] * 1e3 / a_launches
            conv_bytes = (384 + 1024) * n_mbs * args.streams
            out["argb"] = {"value": pics_per_step * n_mbs * side_steps / argb["elapsed"], "unit": "macroblocks/s", "fps": pics_per_step * side_steps / argb["elapsed"],
                           "ms_per_step": argb["elapsed"] * 1e3 / side_steps, "steps": side_steps, "format": "BGRA (the reference's ARGB word, h264bsdConvertToBGRA)",
                           "alg_bytes_per_mb": alg_per_mb + 1024,
                           "whole_path_GBs": (alg_bytes_stream + 1024 * n_mbs * n_pics) * args.streams * side_steps / (argb["dev_ms"] * 1e-3) / 1e9,
                           "conversion": {"ticks_hosted_by_the_next_ticks_k_frame_dbk": n_pics * side_steps - argb["conv_n"], "ticks_followed_by_a_k_convert_tiles_launch": argb["conv_n"],
                                          "hosted_pictures_checked_against_the_reference": argb["hosted_checked"]},
                           "k_convert": {"avg_launch_us": conv_us, "launches": argb["conv_n"], "alg_bytes_per_launch": conv_bytes,
                                         "achieved_GBs": conv_bytes / (conv_us * 1e-6) / 1e9, "frac": conv_bytes / (conv_us * 1e-6) / 1e9 / HBM_PEAK_GBS}}
        if per_gpu is not None:
            out["per_gpu"] = {"value": per_gpu, "unit": "macroblocks/s", "note": "lock-step variant, each rank's own clock over the timed steps"}
        if desync is not None:
            # the same algorithmic bytes per lap as a lock-step step: fraction of the HBM peak and of the lock-step value
            for key, d in desync.items():
                if isinstance(d, dict):
                    d["whole_path_GBs"] = alg_bytes_stream * args.streams * world / (d["ms_per_step"] * 1e-3) / 1e9
                    d["whole_path_frac"] = d["whole_path_GBs"] / HBM_PEAK_GBS / world
                    d["whole_path_frac_moved"] = d["whole_path_frac"] * moved_per_mb / alg_per_mb      # (the copies that elision leaves out do not move)
                    d["vs_lock_step"] = d["value"] / (mbs / elapsed)
            out["desynchronised"] = desync
        for key, leg in e2e.items():
            out[key] = leg
        out["device_errors"] = h264bsd_amd.device_errors()
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(data)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
