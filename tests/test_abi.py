"""The C-ABI library loads on a machine without a GPU and exports every symbol
include/cerberus_mfp.h declares; compute entry points refuse to run without a
device (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

from cerberus_amd import abi, lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "cerberus_mfp.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(cmfp_[a-z0-9_]+)\s*\(", src)
    return sorted(set(n for n in names if not n.endswith("_fn")))


def test_every_declared_symbol_is_exported_and_bound():
    L = lib.load()
    declared = header_functions()
    bound = sorted(n for n, _, _ in lib.SYMBOLS)
    assert declared == bound, (set(declared) ^ set(bound))
    for n in declared:
        assert getattr(L, n) is not None


def test_struct_sizes_match_header_layout(tmp_path):
    """sizeof / offsetof of every POD of include/cerberus_mfp.h as gcc lays them out against the ctypes mirror."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "sz.c"
    src.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "cerberus_mfp.h"\n'
        "int main(void){printf(\"%zu %zu %zu %zu %zu %zu %zu %zu\\n\", sizeof(cmfp_geometry), sizeof(cmfp_params),"
        " sizeof(cmfp_state_desc), sizeof(cmfp_action_desc), offsetof(cmfp_params, n0),"
        " offsetof(cmfp_state_desc, effective_zero), offsetof(cmfp_state_desc, bc),"
        " offsetof(cmfp_action_desc, cross_sections));return 0;}\n")
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    want = [C.sizeof(abi.Geometry), C.sizeof(abi.Params), C.sizeof(abi.StateDesc), C.sizeof(abi.ActionDesc),
            abi.Params.n0.offset, abi.StateDesc.effective_zero.offset, abi.StateDesc.bc.offset,
            abi.ActionDesc.cross_sections.offset]
    assert got == want


def test_version_and_pure_host_entry_points():
    L = lib.load()
    assert b"gfx950" in L.cmfp_version()
    assert L.cmfp_compute_new_dt(1.0, 0.5) == pytest.approx(0.55)   # 1.1 x growth cap, MFP_timestep.cpp:89-92
    assert L.cmfp_compute_new_dt(0.4, 0.5) == 0.4


def test_no_cpu_fallback_without_device():
    L = lib.load()
    if L.cmfp_device_count() > 0:
        pytest.skip("a GPU is present")
    from cerberus_amd import config, problems
    from cerberus_amd.mfp import MFP
    with pytest.raises(lib.CmfpError) as e:
        MFP(config.Problem(problems.kelvin_helmholtz_2d(16)))
    assert e.value.code == abi.ERR_NODEVICE
    assert L.cmfp_leaf_reconstruct(1, 0, 4, None, None, None, None, None) == abi.ERR_NODEVICE


def test_product_does_not_reference_the_oracle():
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "cerberus_amd")):
        if "build" in dirpath or "__pycache__" in dirpath:
            continue
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".H", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                if re.search(r"(from|import)\s+oracle|oracle/|libmfp_oracle", txt):
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad
