"""states.<field>.dynamic = {...}: components that calc_primitives overrides with their initialisation function
(MFP_eulerian.cpp:403-423; read for field states only, MFP_field.cpp:175-221).  Parity unpinned: no input of the
reference on the built path uses the key (Exec/testing/HartmannFlow does, with the BraginskiiCTU action), so the
oracle restates the code and the tests below pin what the override means."""
import numpy as np
import pytest

from cerberus_amd import abi, config, problems


def _zB(d):
    return 0.3 * np.sin(d["x"]) * np.cos(d["y"]) * (1.0 + 0.5 * d["t"])


_zB.uses_time = True


def _problem(scheme="strang", dynamic=True, n=24, fp_mode=abi.FP_STRICT, time_dependent=True):
    cfg = problems.orszag_tang_two_fluid_2d(n)
    cfg["time_integration_scheme"] = scheme
    fn = _zB if time_dependent else (lambda d: 0.3 * np.sin(d["x"]) * np.cos(d["y"]))
    cfg["states"]["field"]["value"]["z_B"] = fn
    if dynamic:
        cfg["states"]["field"]["dynamic"] = ["z_B"]
    return config.Problem(cfg, fp_mode=fp_mode)


def test_dynamic_entries_of_the_configuration():
    P = _problem()
    fs = P.state_names.index("field")
    assert list(P.dynamic_entries(fs)) == [5] and P.dynamic_depends_on_time(fs)
    assert all(not P.dynamic_entries(s) for s in range(3) if s != fs)
    (comp, tab), = P.dynamic_tables(fs, 4, 0.25)
    assert comp == 5 and tab.shape == (1, 24 + 8, 24 + 8)
    # ghost layers hold the function at the ghost-cell centres (x = prob_lo + (i + 0.5) dx, i from -4)
    dx = P.dx[0]
    x = P.prob_lo[0] + (np.arange(-4, 28) + 0.5) * dx
    assert np.allclose(tab[0], 0.3 * np.sin(x)[None, :] * np.cos(x)[:, None] * 1.125, rtol=1e-14)
    # a hydro state's 'dynamic' key is not read by the reference (HydroState::set_udf): ignored
    cfg = problems.orszag_tang_two_fluid_2d(8)
    cfg["states"]["ion"]["dynamic"] = ["rho"]
    assert not config.Problem(cfg).dynamic_entries(config.Problem(cfg).state_names.index("ion"))


def test_oracle_override_acts_on_the_fluxes_not_on_the_state(oracle):
    """euler integrator, one step: the run with dynamic = {z_B} changes every component by what a run whose z_B *is*
    the function changes it by (same primitives, same fluxes), while its own z_B keeps its initial values + dU."""
    Pa = _problem("euler", time_dependent=False)
    Pb = _problem("euler", dynamic=False, time_dependent=False)
    fs = Pa.state_names.index("field")
    a, b = oracle.OracleSim(Pa), oracle.OracleSim(Pb)
    Fa = a.get_state(fs).copy()
    Fa[5] = 0.05                      # the state's own z_B differs from the function
    a.set_state(fs, Fa)
    olda = [a.get_state(s).copy() for s in range(3)]
    oldb = [b.get_state(s).copy() for s in range(3)]
    dt = 0.5 * b.est_time_step()
    a.advance(dt)
    b.advance(dt)
    for s in range(3):
        da, db = a.get_state(s) - olda[s], b.get_state(s) - oldb[s]
        scale = np.abs(oldb[s]).max()
        if s == fs:
            # the sources read the conserved copy that calc_primitives wrote the override into only in actions that
            # come after the field's CTU action in the same pass ("plasma" does): all components agree
            assert np.abs(da - db).max() <= 1e-13 * scale
        else:
            assert np.abs(da - db).max() <= 1e-13 * scale, s
    assert np.abs(a.get_state(fs)[5] - (0.05 + (b.get_state(fs)[5] - oldb[fs][5]))).max() <= 1e-13


def test_oracle_rk2_takes_the_override_into_the_state(oracle):
    """RK2: the second stage's U holds the override, and new = 0.5 old + 0.5 (U + dU) (MFP_advance.cpp:106-127):
    half of (function - state) enters the state per step."""
    P = _problem("RK2", time_dependent=False)
    fs = P.state_names.index("field")
    a = oracle.OracleSim(P)
    F = a.get_state(fs).copy()
    fn = F[5].copy()
    F[5] = 0.0
    a.set_state(fs, F)
    a.advance(1e-6)                   # dU ~ 1e-6: the jump is the override
    assert np.allclose(a.get_state(fs)[5], 0.5 * fn, atol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("scheme", ["strang", "RK2", "euler"])
def test_dynamic_component_parity(oracle, scheme):
    """HIP path (stage-by-stage kernels) against the oracle with a function of (x, y, t): tabulated at time + dt for
    the Strang flux stage, at time for both stages of euler / RK2 (MFP_advance.cpp:72-104,208-217)."""
    from cerberus_amd.mfp import MFP
    for mode, tol in ((abi.FP_STRICT, 0.0), (abi.FP_FAST, 1e-12)):
        P = _problem(scheme, fp_mode=mode, time_dependent=True)
        ref = oracle.OracleSim(P)
        m = MFP(P)
        m.init_data()
        dts_ref, dts = ref.run(4), m.run(4)
        assert m.last_advance_path() == 0          # stage by stage
        if mode == abi.FP_STRICT:
            assert dts == dts_ref
        for s in range(3):
            a, b = m.download(s), ref.get_state(s)
            if tol == 0.0:
                assert np.array_equal(a, b), (scheme, P.state_names[s])
            else:
                from helpers import rel_linf
                assert rel_linf(a, b) <= tol, (scheme, P.state_names[s])
        m.close()


@pytest.mark.gpu
def test_dynamic_refusals_and_removal(oracle):
    from cerberus_amd import lib
    from cerberus_amd.mfp import MFP
    import ctypes as C
    P = _problem(dynamic=False)
    m = MFP(P)
    m.init_data()
    ion = P.state_names.index("ion")
    buf = np.zeros(m.device_view(ion)[2][0] * m.device_view(ion)[2][1])
    assert m.L.cmfp_state_set_dynamic(m.h, ion, 0, buf.ctypes.data_as(lib.dp)) == abi.ERR_INVALID   # hydro state
    fs = P.state_names.index("field")
    assert m.L.cmfp_state_set_dynamic(m.h, fs, 10, buf.ctypes.data_as(lib.dp)) == abi.ERR_INVALID   # bad component
    # an entry, then its removal: the fused kernels are back and the run equals the one that never had it
    ext = m.device_view(fs)[2]
    tab = np.full(ext[0] * ext[1], 0.2)
    m._chk(m.L.cmfp_state_set_dynamic(m.h, fs, 5, tab.ctypes.data_as(lib.dp)))
    m.run(1)
    assert m.last_advance_path() == 0
    m._chk(m.L.cmfp_state_set_dynamic(m.h, fs, 5, None))
    m.init_data()
    m.time, m.dt, m.step = 0.0, None, 0
    m.run(3)
    assert m.last_advance_path() == 1
    ref = oracle.OracleSim(P)
    ref.run(3)
    for s in range(3):
        assert np.array_equal(m.download(s), ref.get_state(s))
    m.close()
