"""Seed-fixed, time-bounded slices of the randomised comparison harnesses of tests/fuzz/ inside the pytest
suite (each harness compares the HIP path with the CPU oracle, brute force, HiGHS or a second kernel on random
inputs, stops at the first difference and prints the inputs).  One child process per harness, one at a time;
the long runs stay `python tests/fuzz/<script> SEED SECONDS`."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

# (script, seed, seconds, extra args, text its summary line must contain)
SLICES = [
    ("fuzz_probe.py", 7001, 10, [], "ok"),
    ("fuzz_sketch.py", 7002, 8, [], "ok"),
    ("fuzz_full_path.py", 7003, 14, [], "fuzz ok"),
    ("fuzz_vs_highs.py", 7004, 12, [], "ok"),
    ("fuzz_dp_kernels.py", 7005, 10, ["2", "257"], "ok"),
    ("fuzz_dp_blocks.py", 7006, 12, ["2", "257"], "ok"),
]


@pytest.mark.parametrize("script,seed,seconds,extra,needle", SLICES, ids=[s[0] for s in SLICES])
def test_fuzz_slice(oracle, script, seed, seconds, extra, needle):
    import __graft_entry__
    __graft_entry__.ensure_built()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz", script), str(seed), str(seconds)] + extra,
                       capture_output=True, text=True, timeout=seconds + 240, cwd=ROOT)
    tail = (r.stdout + r.stderr)[-1500:]
    assert r.returncode == 0, tail
    assert needle in r.stdout.lower() or needle in r.stdout, tail
    assert "FAIL" not in r.stdout, tail
