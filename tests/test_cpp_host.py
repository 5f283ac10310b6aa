"""The C++ host mirror (include/firework.hpp): compiles everywhere, runs the reference's stress_test
parameters on the GPU and must land on the count the emission arithmetic predicts."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples"), "-s"])
    return os.path.join(ROOT, "examples", "stress_test")


def test_cpp_example_builds_and_fails_loudly_without_gpu():
    exe = build()
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 1 and "no CPU fallback" in r.stderr


@pytest.mark.gpu
def test_cpp_stress_test_counts():
    exe = build()
    out = subprocess.run([exe, "160000", "300"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    n = int(re.search(r"Particles: (\d+)", out.stdout).group(1))
    assert n in (157333, 157334, 157335)  # examples/stress_test.rs load: 157 334 per 1 s cycle
