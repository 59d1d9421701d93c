"""Host-side logic that mirrors the reference's set-up code: BC tables, sorted
state/action order, reference scaling, block decomposition."""
import math

import numpy as np
import pytest

from cerberus_amd import abi, bc, config, problems
from cerberus_amd.mfp import block_grid, coords_rank, rank_coords

INT, FO, EVEN, ODD = abi.BC_INT_DIR, abi.BC_FOEXTRAP, abi.BC_REFLECT_EVEN, abi.BC_REFLECT_ODD


def test_hydro_bc_tables():
    # MFP_bc.cpp:11-31,43-89: slipwall in y: scalars even, normal velocity (y) odd, tangential even
    t = bc.hydro_bc({"y": {"lo": {"fill_hydro_bc": "slipwall"}, "hi": {"fill_hydro_bc": "noslipwall"}}}, 2, [1, 0, 0])
    assert [t[c][0][0] for c in range(5)] == [INT] * 5          # periodic x overrides (MFP_hydro.cpp:454-459)
    assert [t[c][1][0] for c in range(5)] == [EVEN, EVEN, ODD, EVEN, EVEN]
    assert [t[c][1][1] for c in range(5)] == [EVEN, ODD, ODD, ODD, EVEN]   # noslipwall: tangential odd too
    # default is outflow -> foextrap (MFP_hydro.cpp:342-345)
    t = bc.hydro_bc({}, 1, [0, 0, 0])
    assert [t[c][0][s] for c in range(5) for s in range(2)] == [FO] * 10
    assert t[5][0][0] == abi.BC_BOGUS


def test_field_bc_tables():
    # MFP_bc.cpp:99-109: symmetry: normal D odd, tangential D even, normal B even, tangential B odd
    d = {"x": {"lo": {"fill_D_bc": "symmetry", "fill_B_bc": "symmetry"},
               "hi": {"fill_D_bc": "asymmetry", "fill_B_bc": "asymmetry", "fill_phi_bc": "symmetry"}}}
    t = bc.field_bc(d, 1, [0, 0, 0])
    assert [t[c][0][0] for c in range(6)] == [ODD, EVEN, EVEN, EVEN, ODD, ODD]
    assert [t[c][0][1] for c in range(6)] == [EVEN, ODD, ODD, ODD, EVEN, EVEN]
    assert t[6][0][1] == EVEN and t[6][0][0] == FO and t[9][0][0] == FO


def test_sorted_order_and_action_species_order():
    P = config.Problem(problems.orszag_tang_two_fluid_2d(16))
    assert P.state_names == ["electron", "field", "ion"]           # get_sorted_keys, MFP_config.cpp:102
    assert P.action_names == ["em_fluxes", "hydro_fluxes", "plasma"]
    ad = P.action_descs()
    assert [ad[1].states[i] for i in range(ad[1].n_states)] == [2, 0]       # CTU{ion, electron}
    assert [ad[2].states[i] for i in range(ad[2].n_states)] == [2, 0, 1]    # species in list order, field last
    sd = P.state_descs()
    assert sd[1].kind == abi.STATE_FIELD and sd[1].reconstruction == abi.RECON["O6"]
    assert sd[0].mass == pytest.approx(1 / 25) and sd[0].charge == -1.0


def test_update_ref():
    # MFP_config.cpp:435-438: Debye = skin_depth / c ; Larmor = sqrt(beta/2) skin_depth
    r = config.update_ref({"lightspeed": 20.0, "skin_depth": 0.2, "beta": 1.0})
    assert r["Debye"] == 0.2 / 20.0 and r["Larmor"] == math.sqrt(0.5) * 0.2
    # :431-432: Larmor & Debye given
    r = config.update_ref({"lightspeed": 10.0, "Larmor": 0.3, "Debye": 0.02})
    assert r["skin_depth"] == 0.2 and r["beta"] == pytest.approx(2 * (0.3 / 0.2) ** 2)
    with pytest.raises(ValueError):
        config.update_ref({"skin_depth": 1.0})


def test_initial_data_prim2cons():
    P = config.Problem(problems.kelvin_helmholtz_2d(8))
    U = P.initial_data(0)
    assert U.shape == (5, 1, 8, 8)
    rho, mx, E = U[0], U[1], U[4]
    u = mx / rho
    assert set(np.unique(rho)) == {1.0, 2.0}
    np.testing.assert_allclose(E, 2.5 / 0.4 + 0.5 * rho * (u * u + (U[2] / rho) ** 2), rtol=1e-15)
    # cell centres: prob_lo + (i + 0.5) dx  (MFP_hydro.cpp:531-536)
    assert P.cell_centres()["x"][0, 0, 0] == 0.5 / 8


def test_block_grid_and_rank_maps():
    assert block_grid(2, 1) == [1, 1, 1]
    assert block_grid(2, 2) == [1, 2, 1]
    assert block_grid(2, 4) == [2, 2, 1]
    assert block_grid(2, 8) == [2, 4, 1]
    assert block_grid(3, 2) == [1, 1, 2]
    assert block_grid(3, 4) == [1, 2, 2]
    assert block_grid(3, 8) == [2, 2, 2]
    g = block_grid(2, 8)
    for r in range(8):
        assert coords_rank(rank_coords(r, g), g) == r
    with pytest.raises(ValueError):
        block_grid(2, 3)


def test_unsupported_options_fail_loudly():
    cfg = problems.orszag_tang_two_fluid_2d(16)
    cfg["states"]["field"]["flux"] = "HLLE"
    with pytest.raises(NotImplementedError):
        config.Problem(cfg).state_descs()
    cfg = problems.kelvin_helmholtz_2d(16)
    cfg["states"]["fluid"]["gas"]["mass"] = [1.0, 2.0]            # a mixture: built for the transport (test_mixture.py)
    assert config.Problem(cfg).state_descs()[0].mixture == 1
    cfg["states"]["fluid"]["gas"]["charge"] = [0.0, 1.0]          # species of different charge: read by the sources
    assert list(config.Problem(cfg).state_descs()[0].species_charge)[:2] == [0.0, 1.0]
    cfg["states"]["fluid"]["gas"]["charge"] = [0.0, 0.0, 0.0]     # MFP_thermally_perfect_gas.cpp:36-39
    with pytest.raises(ValueError):
        config.Problem(cfg).state_descs()
