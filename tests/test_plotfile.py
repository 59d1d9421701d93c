"""AMReX plotfile writer: layout parsed back by a reader restated from the
format, and, where the reference tree is present (this container only), by the reference's own
vis/get_boxlib.py."""
import importlib.util
import json
import os
import re

import numpy as np
import pytest

from cerberus_amd import config, plotfile, problems

REF_READER = "/root/reference/vis/get_boxlib.py"


def _case():
    cfg = problems.orszag_tang_two_fluid_2d(48)
    cfg["n_cell"] = [48, 40]
    P = config.Problem(cfg)
    return P, [P.initial_data(s) for s in range(3)]


def _read_back(path, name):
    """minimal reader of the layout: Header names, Cell_H boxes + FabOnDisk, FAB payloads"""
    with open(os.path.join(path, "Header")) as f:
        assert f.readline().strip() == "HyperCLaw-V1.1"
        n = int(f.readline())
        names = [f.readline().rstrip() for _ in range(n)]
        dim = int(f.readline())
        time = float(f.readline())
        assert int(f.readline()) == 0
        f.readline(); f.readline(); f.readline()
        dom = [int(v) for v in re.findall(r"-?\d+", f.readline())]
    with open(os.path.join(path, "Level_0", "Cell_H")) as f:
        assert [int(f.readline()) for _ in range(4)] == [1, 1, n, 0]
        nb = int(re.findall(r"\d+", f.readline())[0])
        boxes = [[int(v) for v in re.findall(r"-?\d+", f.readline())] for _ in range(nb)]
        assert f.readline().strip() == ")"
        assert int(f.readline()) == nb
        fod = [f.readline().split() for _ in range(nb)]
    hi = dom[dim:2 * dim]
    out = np.full([h + 1 for h in hi][::-1], np.nan)
    c = names.index(name)
    for b, (_, fname, off) in zip(boxes, fod):
        lo_, hi_ = b[:dim], b[dim:2 * dim]
        shape = [hi_[d] - lo_[d] + 1 for d in range(dim)][::-1]
        with open(os.path.join(path, "Level_0", fname), "rb") as f:
            f.seek(int(off))
            assert f.readline().startswith(b"FAB ")
            f.seek(int(np.prod(shape)) * 8 * c, 1)
            a = np.frombuffer(f.read(int(np.prod(shape)) * 8), "<f8").reshape(shape)
        out[tuple(slice(lo_[d], hi_[d] + 1) for d in range(dim))[::-1]] = a
    return names, time, out


@pytest.mark.parametrize("mgs", [None, 16])
def test_plotfile_layout_roundtrip(tmp_path, mgs):
    P, states = _case()
    d = str(tmp_path / "plt00007")
    names = plotfile.write_plotfile(d, P, states, time=0.125, step=7, max_grid_size=mgs)
    assert names == sorted(names) and "cost" in names and "x_D-field" in names and "T-ion" in names
    ion, fld = P.state_index["ion"], P.state_index["field"]
    got_names, time, rho = _read_back(d, "rho-ion")
    assert got_names == names and time == 0.125
    assert np.array_equal(rho, states[ion][0, 0])
    assert np.array_equal(_read_back(d, "y_B-field")[2], states[fld][4, 0])
    # primitive output follows cons2prim (MFP_thermally_perfect_gas.cpp:180-218)
    U = states[ion][:, 0]
    p = (U[4] - 0.5 * (U[1] ** 2 + U[2] ** 2 + U[3] ** 2) / U[0]) * (5.0 / 3.0 - 1.0)
    np.testing.assert_allclose(_read_back(d, "p-ion")[2], p, rtol=1e-13)
    info = json.load(open(os.path.join(d, "info")))
    assert info["num_state"] == 3 and info["state_%d" % ion]["gas"]["mass"] == [1.0]
    assert info["state_%d" % fld]["cons_names"] == plotfile.FIELD_CONS
    assert abs(info["Debye"] - 0.01) < 1e-15 and info["lightspeed"] == 20.0


def test_plotfile_of_a_mixture(tmp_path):
    """tracer_i / alpha_i variables (MFP_hydro.cpp:583-586) and the per-cell mass, charge, gamma, cp, T of a gas of
    several species (get_plot_output :702-717)"""
    cfg = problems.shock_bubble_2d(32, 16, species=3)
    cfg["states"]["fluid"]["gas"]["charge"] = [0.0, 1.0, 2.0]
    P = config.Problem(cfg)
    U = P.initial_data(0)
    d = str(tmp_path / "plt00000")
    names = plotfile.write_plotfile(d, P, [U], time=0.0, step=0)
    for v in ("tracer_0-fluid", "alpha_1-fluid", "mass-fluid", "gamma-fluid", "cp-fluid", "charge-fluid"):
        assert v in names
    assert np.array_equal(_read_back(d, "tracer_1-fluid")[2], U[6, 0])
    a0, a1 = U[5, 0] / U[0, 0], U[6, 0] / U[0, 0]
    al = [a0, a1, 1 - (a0 + a1)]
    m, g, q = [1.0, 2.0, 0.138], [1.4, 1.1, 5.0 / 3.0], [0.0, 1.0, 2.0]
    n = sum(a / mi for a, mi in zip(al, m))
    np.testing.assert_allclose(_read_back(d, "mass-fluid")[2], 1 / n, rtol=1e-14)
    np.testing.assert_allclose(_read_back(d, "charge-fluid")[2], sum(a * qi / mi for a, mi, qi in zip(al, m, q)) / n, rtol=1e-13)
    cp = sum(a * gi / (mi * (gi - 1)) for a, mi, gi in zip(al, m, g))
    cv = sum(a / (mi * (gi - 1)) for a, mi, gi in zip(al, m, g))
    np.testing.assert_allclose(_read_back(d, "gamma-fluid")[2], cp / cv, rtol=1e-14)
    np.testing.assert_allclose(_read_back(d, "cp-fluid")[2], cp, rtol=1e-14)
    np.testing.assert_allclose(_read_back(d, "p-fluid")[2][:, 20:], 1.0, rtol=1e-13)      # ahead of the shock
    info = json.load(open(os.path.join(d, "info")))
    assert info["state_0"]["gas"]["mass"] == m and info["state_0"]["gas"]["charge"] == q


@pytest.mark.skipif(not os.path.exists(REF_READER), reason="reference tree not present (GPU box)")
def test_reference_reader_reads_our_plotfile(tmp_path):
    """The reference's post-processing reader parses the plotfile and returns the data bit for bit."""
    if not hasattr(np, "product"):
        np.product = np.prod          # removed in NumPy 2; the reference's reader still calls it
    spec = importlib.util.spec_from_file_location("get_boxlib", REF_READER)
    gb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gb)
    P, states = _case()
    d = str(tmp_path / "plt00007")
    plotfile.write_plotfile(d, P, states, time=0.125, step=7, max_grid_size=16)
    rh = gb.ReadBoxLib(d)
    assert rh.time == 0.125 and rh.names == P.state_names
    assert rh.data["levels"][0]["num_grids"] == 9
    for var, s, c in (("rho-ion", "ion", 0), ("nrg-electron", "electron", 4), ("z_B-field", "field", 5)):
        x, v = rh.get(var)
        assert np.array_equal(v, states[P.state_index[s]][c, 0].T), var
    assert np.allclose(x[0], P.prob_lo[0] + (np.arange(48) + 0.5) * P.dx[0])
    assert rh.data["hydro_mass"] == [[0.04], [1.0]] and rh.data["hydro_charge"] == [[-1.0], [1.0]]
    rh.close()


@pytest.mark.gpu
def test_plotfile_from_gpu_run(tmp_path):
    from cerberus_amd.mfp import MFP
    P = config.Problem(problems.kelvin_helmholtz_2d(32))
    m = MFP(P)
    m.init_data()
    m.run(3)
    d = str(tmp_path / "plt00003")
    m.write_plotfile(d, max_grid_size=16)
    names, time, rho = _read_back(d, "rho-fluid")
    assert time == m.time and np.array_equal(rho, m.download(0)[0, 0])
    m.close()
