"""CPU-side checks of the product library: it loads, exports every symbol of
include/firework_hip.h, its host count arithmetic is bit-exact, and it refuses to
run without a GPU (no fallback).  No compute calls are made here."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

from bevy_firework_amd import _ffi, system
from bevy_firework_amd import settings as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def f(bits):
    return np.array([bits], dtype=np.uint32).view(np.float32)[0]


def b(x):
    return int(np.asarray(x, dtype=np.float32).view(np.uint32))


def test_library_exports_every_declared_symbol():
    lib = _ffi.load()
    header = open(os.path.join(ROOT, "include", "firework_hip.h")).read()
    product = set(re.findall(r"\b(fw_[a-z0-9_]+)\s*\(", header)) - {"fw_status"}
    # measurement / debugging hooks live in a header of their own: exported, not part of the surface a host binds
    debug = set(re.findall(r"\b(fw_[a-z0-9_]+)\s*\(", open(os.path.join(ROOT, "include", "firework_hip_debug.h")).read()))
    assert debug and not (debug & product) and all(n.startswith(("fw_debug_", "fw_ctx_kernel_timing", "fw_ctx_measure_")) for n in debug), debug
    declared = product | debug
    bound = {name for name, _, _ in _ffi.SYMBOLS}
    assert declared == bound, declared ^ bound
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.fw_abi_version() == 5
    # the Rust `extern "C"` block of INTEGRATION.md lists the same functions (three mirrors of one header: keep them in step)
    rust = set(re.findall(r"pub fn (fw_[a-z0-9_]+)\(", open(os.path.join(ROOT, "INTEGRATION.md")).read()))
    assert rust == product, rust ^ product
    # ... and so does the shim's source form (rust/src/hip/ffi.rs: unverified Rust, but the same list)
    shim = set(re.findall(r"pub fn (fw_[a-z0-9_]+)\(", open(os.path.join(ROOT, "rust", "src", "hip", "ffi.rs")).read()))
    assert shim == product, shim ^ product
    # ... and the library exports no fw_* function the header does not declare
    import subprocess

    exported = set(re.findall(r" T (fw_[a-z0-9_]+)$", subprocess.check_output(["nm", "-D", "--defined-only", _ffi.LIB_PATH], text=True), re.M))
    assert exported == declared, exported ^ declared


def test_struct_sizes_match_header(tmp_path):
    """the ctypes mirrors against the C compiler's layout of include/firework_hip.h"""
    import subprocess

    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include "firework_hip.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu\\n",'
                   "sizeof(fw_particle_settings),sizeof(fw_emission_settings),sizeof(fw_spawner_desc),sizeof(fw_collider),"
                   "sizeof(fw_particle),sizeof(fw_particle_instance));return 0;}\n")
    exe = tmp_path / "sizes"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert got == [C.sizeof(_ffi.ParticleSettings), C.sizeof(_ffi.EmissionSettings), C.sizeof(_ffi.SpawnerDesc),
                   C.sizeof(_ffi.Collider), S.PARTICLE_DTYPE.itemsize, S.INSTANCE_DTYPE.itemsize], got
    assert S.PARTICLE_DTYPE.itemsize == 104 and S.INSTANCE_DTYPE.itemsize == 64


def test_host_emission_count_is_bit_exact_with_golden():
    """fw_compute_emission_count is the arithmetic fw_step's host side uses for Global entries."""
    d = json.load(open(os.path.join(G, "emission_kat.json")))
    total = 0
    for age_b, last_b, n, next_b in d["steps"]:
        gn, gnext = system.compute_emission_count(f(age_b), f(last_b), 3.0, 0.0, 1.0, 23.0)
        assert gn == n and b(gnext) == next_b
        total += gn
    assert total in (22, 23)  # reference src/core.rs:830-833
    w = json.load(open(os.path.join(G, "emission_wrap.json")))
    for case in w["cases"]:
        last = np.float32(0)
        for tpc_b, n, last_b in case["frames"]:
            gn, last = system.compute_emission_count(f(tpc_b), last, case["duration"], case["offset_start"],
                                                     case["offset_end"], case["count"])
            assert gn == n and b(last) == last_b
    nk = json.load(open(os.path.join(G, "nested_count_kat.json")))
    for case in nk["cases"]:
        last = np.float32(np.finfo(np.float32).min)
        for age_b, n, last_b in case["rows"]:
            gn, last = system.compute_emission_count(f(age_b), last, case["lifetime"], case["offset_start"],
                                                     case["offset_end"], case["count"])
            assert gn == n and b(last) == last_b


def test_no_cpu_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(system.FwError) as e:
        system.ParticleSystem()
    assert e.value.status == _ffi.FW_ENODEV


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "bevy_firework_amd")
    for dirpath, _, files in os.walk(pkg):
        for name in files:
            if name.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, name)).read()
                assert "import oracle" not in src and "from oracle" not in src and "fw_oracle" not in src, name
