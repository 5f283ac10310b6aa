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


# (round 5: ... and a fourth time with every type that qualifies -- nothing emits on or from it, no collisions, no instance
# buffer -- on the wave-per-type kernel of small types, fw_k_small.hip, WHATEVER its size: FW_SMALL_MAX huge; the other three
# paths switch that kernel off, so "general" still means the compacting kernels at every size)
# Every GPU test runs three times: with constant-lifetime particle types on the in-place FIFO ring path (whatever their
# size: FW_FIFO_MIN=0), with every eligible particle type -- any lifetime range, no Nested entries in its spawner, no
# collisions -- on an in-place RANGE ring (FW_RANGE_MIN=0; FIFO rings off, so constant lifetimes take it too), and with
# both switched off (everything on the general, compacting path).  The knobs are read when a context is created, so setting
# the environment before the test body is enough.
def pytest_generate_tests(metafunc):
    if metafunc.module.__name__.split(".")[-1] in ("test_gpu_range", "test_gpu_lifecycle", "test_gpu_host_fast", "test_gpu_thresholds", "test_gpu_vardt"):
        return  # (set their own knobs: every test of test_gpu_range is about one path, the other two run at product defaults)
    if metafunc.definition.get_closest_marker("gpu") and "fw_path" in metafunc.fixturenames:
        metafunc.parametrize("fw_path", ["fifo", "range", "general", "small"], indirect=True)


def _four_round_tiles(request) -> bool:
    """Ring launches of fewer than FW_FIFO_SMALL / FW_RANGE_SMALL four-round tiles in all run on one-round tiles (TR = 1) -- what
    the product picks for nearly every test of the suite, whose rings are small; large rings run the four-round kernels
    (TR = 4).  Both instantiations need the suite: every test FUNCTION is assigned one of the two, by a hash of its name (stable
    across runs and paths), so about half of the parity / limits / examples / golden tests exercise the kernels the product
    itself would pick at their size and the other half the ones it picks at BASELINE sizes (the suite used to force
    four-round tiles nearly everywhere).  tests/test_gpu_fifo.py and tests/test_gpu_range.py run both forms of every test of
    theirs; the fuzz and tests/test_gpu_lifecycle.py use the product's choice."""
    import zlib

    if request.module.__name__.split(".")[-1] == "test_gpu_fuzz":
        return False
    return (zlib.crc32(request.node.originalname.encode() if getattr(request.node, "originalname", None) else request.node.name.encode()) & 1) == 0


def _name_bits(request) -> int:
    import zlib

    # (which instantiation a test function runs follows from its NAME -- FW_TEST_ROTATE=n, any integer, deals the variants out
    # differently: a run of the suite with another n covers the combinations the committed one does not; the variant is in every log)
    rot = os.environ.get("FW_TEST_ROTATE", "")
    name = request.node.originalname if getattr(request.node, "originalname", None) else request.node.name
    return zlib.crc32((name + rot).encode())


@pytest.fixture(autouse=True)
def fw_path(request, monkeypatch):
    mode = getattr(request, "param", None)
    monkeypatch.setenv("FW_ENABLE_KNOBS", "1")  # the library reads its A/B switches only with this set (firework_hip_debug.h)
    # Per-frame records and small op tables live in device memory the host writes through the large BAR where the platform maps it
    # (fw_ctx::param_bar), in pinned host memory otherwise: a quarter of the test FUNCTIONS of the path matrix (by the same kind of
    # hash as the tile sizes below) run the pinned form, so both stay under the suite whatever the box offers.
    if mode is not None and (_name_bits(request) >> 1) & 3 == 0:
        monkeypatch.setenv("FW_PARAM_BAR", "0")
    # Round 6: the update of every type leaves scale and colours to its readers (FW_TYPE_DERIVED for all, fw_ctx::derive_all); a
    # quarter of the test functions of the path matrix keeps the form of rounds 3-5 -- the three planes stored unless an instance
    # buffer is attached (FW_DERIVED=1) -- so the stores, fw_k_rederive and the transitions between the two stay under the suite.
    if mode is not None and (_name_bits(request) >> 5) & 3 == 0 and "FW_DERIVED" not in os.environ:
        monkeypatch.setenv("FW_DERIVED", "1")
    if mode in ("fifo", "range", "general"):
        monkeypatch.setenv("FW_SMALL", "0")
    if mode == "small":
        monkeypatch.setenv("FW_FIFO", "0")
        monkeypatch.setenv("FW_RANGE", "0")
        monkeypatch.setenv("FW_SMALL", "1")
        monkeypatch.setenv("FW_SMALL_MIN", "0")
        # (every eligible type whatever its size: on a WAVE each in half of the test functions, a WORKGROUP each -- the kernel's wide
        # role -- for what sustains more than a few hundred particles in the other half)
        if (_name_bits(request) >> 3) & 1:
            monkeypatch.setenv("FW_SMALL_MAX", "2000000000"), monkeypatch.setenv("FW_WIDE_MAX", "0")
        else:
            monkeypatch.setenv("FW_WIDE_MAX", "2000000000"), monkeypatch.setenv("FW_WIDE_MIN", "0")
    if mode == "fifo":
        monkeypatch.setenv("FW_FIFO", "1")
        monkeypatch.setenv("FW_FIFO_MIN", "0")
        monkeypatch.setenv("FW_RANGE", "0")
        if _four_round_tiles(request):
            monkeypatch.setenv("FW_FIFO_SMALL", "0")
    elif mode == "range":
        monkeypatch.setenv("FW_FIFO", "0")
        monkeypatch.setenv("FW_RANGE", "1")
        monkeypatch.setenv("FW_RANGE_MIN", "0")
        if _four_round_tiles(request):
            monkeypatch.setenv("FW_RANGE_SMALL", "0")
    elif mode == "general":
        monkeypatch.setenv("FW_FIFO", "0")
        monkeypatch.setenv("FW_RANGE", "0")
        # (round 6: the threshold forecast -- fw_k_fc_resolve in front of the streaming kernel when dt differs from the previous
        # frame's -- engages from 2048 tiles on in the product; half of the test functions run it at any size)
        if (_name_bits(request) >> 7) & 1:
            monkeypatch.setenv("FW_TF_MIN_TILES", "0")
    if mode is not None:
        # which instantiations this test function ran (the choice depends on the test's name -- it is in the log of
        # a failing test and in the junit properties)
        variant = {k: os.environ.get(k) for k in ("FW_PARAM_BAR", "FW_DERIVED", "FW_FIFO_SMALL", "FW_RANGE_SMALL", "FW_SMALL_MAX", "FW_WIDE_MAX", "FW_TF_MIN_TILES")
                   if os.environ.get(k) is not None}
        request.node.user_properties.append(("fw_variant", f"{mode}: {variant}"))
        print(f"[fw variant] path={mode} {variant}")
    return mode
