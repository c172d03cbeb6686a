"""N>1 plumbing on CPU: two ranks over gloo, each owning its own shard of independent streams (weak
scaling, no data-path collective) — the same sharding + MAX-of-elapsed reduction bench.py uses over RCCL.
Each rank runs the PRODUCT's host parser in capture mode on its streams; nothing here needs a GPU."""
import os
import subprocess
import sys
import textwrap

from conftest import ROOT

WORKER = textwrap.dedent("""
    import os, sys, json, hashlib, time
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    import h264bsd_amd
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    streams_total, per_rank = 4, 2
    mine = list(range(rank * per_rank, (rank + 1) * per_rank))          # stream s -> rank s // per_rank
    data = open(os.path.join(%r, "tests", "golden", "test_640x360.h264"), "rb").read()
    dist.barrier()
    t0 = time.perf_counter()
    pics = 0
    digests = []
    for s in mine:
        jobs, trace, info = h264bsd_amd.capture_stream(data)
        pics += len(jobs)
        digests.append(hashlib.sha256(b"".join(jobs)).hexdigest())
    dist.barrier()
    elapsed = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    total = torch.tensor([pics], dtype=torch.int64)
    dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    dist.all_reduce(total, op=dist.ReduceOp.SUM)
    gathered = [None] * world
    dist.all_gather_object(gathered, digests)
    if rank == 0:
        print(json.dumps({"pics": int(total.item()), "elapsed": float(elapsed.item()),
                          "distinct_job_streams": len({d for g in gathered for d in g}), "world": world}))
    dist.destroy_process_group()
""") % (ROOT, ROOT)


def test_two_ranks_shard_streams_over_gloo(tmp_path, built):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29631", str(script)],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["world"] == 2 and res["pics"] == 4 * 73          # every rank decoded its own 2 streams
    assert res["distinct_job_streams"] == 1                        # identical copies -> identical frame jobs


import pytest  # noqa: E402


@pytest.mark.gpu
def test_gpu_bench_two_ranks_on_one_box(built):
    """the N > 1 path of bench.py before an 8-GPU node runs it: two ranks under torch.distributed.run (on one GPU they
    share the device and reduce over gloo; on a multi-GPU box each takes its own device over RCCL), 32 streams each,
    verified on device; the line must carry the node total AND the per-GPU values"""
    import json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29641", os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--streams", "32", "--steps", "1", "--warmup", "0", "--ramp-seconds", "0",
                          "--no-staggered", "--no-desync", "--no-argb", "--no-full-copies-variant"],
                         capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert res["n_gpus"] == 2 and res["config"]["pictures_per_step"] == 2 * 32 * 73
    assert len(res["per_gpu"]["value"]) == 2 and all(v > 0 for v in res["per_gpu"]["value"])
    assert abs(res["value"] - 2 * 32 * 73 * 8160 * 1000 / res["ms_per_step"]) < 1e-3 * res["value"]
    assert res["device_errors"] == 0 and "cpu_baseline" not in res
    # the end-to-end legs run on both ranks at once and the line carries them: the node figure, every rank's
    # own, and parser pools that together do not oversubscribe what the container may use
    import h264bsd_amd.capi  # noqa: F401
    for key in ("end_to_end", "end_to_end_host_output"):
        leg = res[key]
        assert leg["n_gpus"] == 2 and leg["streams"] == 64 and len(leg["per_gpu"]["fps"]) == 2 and all(v > 0 for v in leg["per_gpu"]["fps"])
        assert leg["fps"] <= sum(leg["per_gpu"]["fps"]) * 1.001
        threads = leg["per_gpu"]["parser_threads"]
        quota = leg["cpu_quota"] or leg["host_cores"]
        assert sum(threads) <= max(2, int(1.25 * quota) + 2), (threads, quota)
    assert res["end_to_end_host_output"]["d2h_bytes_per_picture"] == 8160 * 384 and res["end_to_end"]["d2h_bytes_per_picture"] == 0


@pytest.mark.gpu
def test_gpu_bench_starts_its_own_ranks(built):
    """`python bench.py --gpus 2` without a launcher (how the driver calls it): bench.py starts the two ranks itself and the
    line says n_gpus 2; a rank count that does not match --gpus is refused instead of silently measuring one GPU"""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--streams", "32", "--steps", "1", "--warmup", "0",
                          "--ramp-seconds", "0", "--no-staggered", "--no-desync", "--no-argb", "--no-full-copies-variant", "--no-end-to-end"],
                         capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, "\n".join(l for l in out.stderr.splitlines() if "socket.cpp" not in l and "amdgpu.ids" not in l)[-6000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert res["n_gpus"] == 2 and len(res["per_gpu"]["value"]) == 2
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--streams", "8", "--steps", "1"],
                         capture_output=True, text=True, timeout=300, env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), cwd=ROOT)
    assert bad.returncode != 0 and "refusing" in bad.stderr + bad.stdout
