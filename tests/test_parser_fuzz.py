"""Robustness of the host parser: arbitrarily damaged streams must end in error codes and concealed pictures, never
in a crash or a hang.  Runs in a child process so that a memory fault shows up as a failed test instead of killing
pytest.  (tests/fuzz_asan/ is the same idea under ASan/UBSan, with every frame job also rendered by the oracle.)"""
import subprocess
import sys
import textwrap

from conftest import ROOT

CHILD = textwrap.dedent("""
    import ctypes, sys
    sys.path.insert(0, %r)
    import numpy as np
    import h264bsd_amd
    data = np.frombuffer(open(%r + "/tests/golden/test_640x360.h264", "rb").read(), dtype=np.uint8)
    rng = np.random.default_rng(int(sys.argv[1]))
    n_cases, pics, errors = 0, 0, 0
    for case in range(int(sys.argv[2])):
        d = data.copy()
        kind = case %% 4
        if kind == 0:                                   # random byte flips
            idx = rng.integers(0, d.size, rng.integers(1, 40)); d[idx] = rng.integers(0, 256, idx.size)
        elif kind == 1:                                 # truncation
            d = d[: int(rng.integers(10, d.size))]
        elif kind == 2:                                 # a burst of garbage
            s = int(rng.integers(0, d.size - 600)); d[s:s + 500] = rng.integers(0, 256, 500)
        else:                                           # bit flips inside the first slices
            idx = rng.integers(30, 4000, 8); d[idx] ^= (1 << rng.integers(0, 8, idx.size)).astype(np.uint8)
        jobs = []
        dec = h264bsd_amd.Decoder(capture=jobs.append)
        buf = ctypes.create_string_buffer(d.tobytes(), d.size)
        base, off, guard, stuck = ctypes.addressof(buf), 0, 0, 0
        while off < d.size and guard < 5000:
            guard += 1
            r, rb = dec.decode(base + off, d.size - off)
            assert 0 <= r <= 5 and rb <= d.size - off
            if r >= 3:
                errors += 1
            if rb == 0:
                stuck += 1
                if stuck > 3:
                    off += 1                          # a caller's resync step after repeated "call me again"
                    stuck = 0
            else:
                stuck = 0
            off += rb
        pics += len(jobs)
        dec.close()
        n_cases += 1
    print("OK", n_cases, pics, errors)
""") % (ROOT, ROOT)


def test_damaged_streams_never_crash(tmp_path, built):
    script = tmp_path / "fuzz_child.py"
    script.write_text(CHILD)
    for seed in (1, 2):
        out = subprocess.run([sys.executable, str(script), str(seed), "60"], capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, f"seed {seed}: rc={out.returncode}\n{out.stderr[-1500:]}"
        tag, n, pics, errors = out.stdout.split()[-4:]
        assert tag == "OK" and int(n) == 60 and int(errors) > 0
