import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


def _has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


# Every GPU test runs three times: with constant-lifetime particle types on the in-place FIFO ring path (whatever their
# size: FW_FIFO_MIN=0), with every eligible particle type -- any lifetime range, no Nested entries in its spawner, no
# collisions -- on an in-place RANGE ring (FW_RANGE_MIN=0; FIFO rings off, so constant lifetimes take it too), and with
# both switched off (everything on the general, compacting path).  The knobs are read when a context is created, so setting
# the environment before the test body is enough.
def pytest_generate_tests(metafunc):
    if metafunc.module.__name__.split(".")[-1] in ("test_gpu_range", "test_gpu_lifecycle"):
        return  # (set their own knobs: every test of test_gpu_range is about one path, test_gpu_lifecycle runs without any knob)
    if metafunc.definition.get_closest_marker("gpu") and "fw_path" in metafunc.fixturenames:
        metafunc.parametrize("fw_path", ["fifo", "range", "general"], indirect=True)


@pytest.fixture(autouse=True)
def fw_path(request, monkeypatch):
    mode = getattr(request, "param", None)
    monkeypatch.setenv("FW_ENABLE_KNOBS", "1")  # the library reads its A/B switches only with this set (firework_hip_debug.h)
    if mode == "fifo":
        monkeypatch.setenv("FW_FIFO", "1")
        monkeypatch.setenv("FW_FIFO_MIN", "0")
        monkeypatch.setenv("FW_RANGE", "0")
        # (FIFO launches of fewer than FW_FIFO_SMALL four-round tiles use one-round tiles: nearly every test of the suite would.
        # The suite keeps the four-round tiles -- what large rings run -- except in tests/test_gpu_fifo.py, which runs both, the
        # collision tests -- colliding launches always use one-round tiles -- the fuzz's default environments and the product-default
        # tests of tests/test_gpu_configs.py)
        if request.module.__name__.split(".")[-1] != "test_gpu_fuzz":
            monkeypatch.setenv("FW_FIFO_SMALL", "0")
    elif mode == "range":
        monkeypatch.setenv("FW_FIFO", "0")
        monkeypatch.setenv("FW_RANGE", "1")
        monkeypatch.setenv("FW_RANGE_MIN", "0")
        # (as for FIFO rings: small range launches use one-round tiles; the suite keeps the four-round tiles except in
        # tests/test_gpu_range.py, which runs both, the fuzz's default environments and the product-default tests)
        if request.module.__name__.split(".")[-1] != "test_gpu_fuzz":
            monkeypatch.setenv("FW_RANGE_SMALL", "0")
    elif mode == "general":
        monkeypatch.setenv("FW_FIFO", "0")
        monkeypatch.setenv("FW_RANGE", "0")
    return mode
