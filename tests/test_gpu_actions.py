"""Action-API level parity (one call per virtual of class Action), ghost fill,
dt limits, option coverage and error behaviour on the GPU against the oracle."""
import numpy as np
import pytest

from cerberus_amd import abi, config, lib, problems
from helpers import rel_linf, small_cases

pytestmark = pytest.mark.gpu


def _gpu(P, **kw):
    from cerberus_amd.mfp import MFP
    m = MFP(P, **kw)
    m.init_data()
    return m


@pytest.mark.parametrize("case", range(5))
def test_action_dU_and_dt_limits(oracle, case):
    """get_data + calc_time_derivative + calc_spatial_derivative of every action,
    get_allowed_time_step of every state and action: bit-exact in STRICT mode."""
    name, cfg, _ = small_cases()[case]
    P = config.Problem(cfg, fp_mode=abi.FP_STRICT)
    ref = oracle.OracleSim(P)
    m = _gpu(P)
    dt = ref.est_time_step()
    v = m.est_time_step()
    assert v == dt
    for s in range(m.n_states):
        assert m.state_allowed_dt(s) == ref.state_allowed_dt(s)
    for a in range(m.n_actions):
        assert m.action_allowed_dt(a) == ref.action_allowed_dt(a)
        m.update_begin()
        m.action_get_data(a)
        m.action_calc_time_derivative(a, 0.5 * dt)
        m.action_calc_spatial_derivative(a, dt)
        ad = P.action_descs()[a]
        for i in range(ad.n_states):
            s = ad.states[i]
            if ad.kind == abi.ACTION["CTU"]:
                exp = ref.action_dU(a, dt, s)
            else:
                exp = ref.action_dU(a, 0.5 * dt, s)
            got = m.update_download(s, 1)
            assert np.array_equal(got, exp), (name, P.action_names[a], P.state_names[s], rel_linf(got, exp))
    m.close()


@pytest.mark.parametrize("case", [3, 4, 0])
def test_fill_patch_matches_oracle(oracle, case):
    import ctypes as C
    name, cfg, _ = small_cases()[case]
    P = config.Problem(cfg)
    ref = oracle.OracleSim(P)
    m = _gpu(P)
    for s in range(m.n_states):
        m.fill_patch(s)
        ptr, ng, ext, nc = m.device_view(s)
        n = nc * ext[0] * ext[1] * ext[2]
        host = np.empty(n)
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        assert hip.hipDeviceSynchronize() == 0
        assert hip.hipMemcpy(host.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), C.c_size_t(8 * n), 2) == 0
        A = host.reshape(nc, ext[2], ext[1], ext[0])
        G = ref.fill_patch(s)                       # domain grown by the state's own num_grow
        gs = m.L.cmfp_state_ngrow(m.h, s)
        off = ng - gs
        sl = tuple(slice(off, e - off) if d < P.dim else slice(None) for d, e in zip((2, 1, 0), (ext[2], ext[1], ext[0])))
        assert np.array_equal(A[(slice(None),) + sl], G), (name, P.state_names[s])
    m.close()


OPTION_CASES = [
    ("implicit", lambda: problems.orszag_tang_two_fluid_2d(32, solver="implicit"), 1e-12, 1e-12),
    ("lorentz", lambda: problems.orszag_tang_two_fluid_2d(32, source="Lorentz"), 0.0, 1e-12),
]


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("idx", range(len(OPTION_CASES)))
def test_source_variants(oracle, idx, fused):
    name, mk, tol_strict, tol_fast = OPTION_CASES[idx]
    for mode, tol in ((abi.FP_STRICT, tol_strict), (abi.FP_FAST, tol_fast)):
        P = config.Problem(mk(), fp_mode=mode)
        ref = oracle.OracleSim(P)
        ref.run(6)
        m = _gpu(P, fused=fused)
        m.run(6)
        assert m.last_advance_fused() == fused
        for s in range(m.n_states):
            assert rel_linf(m.download(s), ref.get_state(s)) <= tol, (name, mode, P.state_names[s])
        m.close()


def _kh(recon, flux, scheme="strang", ctu=True, thr=None, n=32):
    cfg = problems.kelvin_helmholtz_2d(n, flux=flux, recon=recon)
    cfg["time_integration_scheme"] = scheme
    cfg["actions"]["fluxes"]["corner_transport"] = ctu
    if thr:
        cfg["states"]["fluid"]["shock_detector"] = {"name": "pressure_jump_detector", "threshold": thr}
    return cfg


@pytest.mark.parametrize("recon", ["constant", "minmod", "vanLeer", "MC", "centre", "O6", "WENO"])
@pytest.mark.parametrize("flux", ["HLLC", "HLLE"])
def test_hydro_option_matrix(oracle, recon, flux):
    P = config.Problem(_kh(recon, flux), fp_mode=abi.FP_STRICT)
    ref = oracle.OracleSim(P)
    ref.run(5)
    for fused in (False, True, 2, 3):
        m = _gpu(P, fused=fused)
        m.run(5)
        tiled = recon in ("minmod", "vanLeer", "MC", "centre")   # the rest: gather kernels only
        assert m.last_advance_path() == (min(int(fused), 2) if fused == 2 or not fused else (1 if tiled else 2))
        assert np.array_equal(m.download(0), ref.get_state(0)), (recon, flux, fused)
        m.close()


@pytest.mark.parametrize("scheme,ctu,thr", [("euler", True, None), ("RK2", True, None), ("strang", False, None),
                                            ("strang", True, 0.3), ("RK2", False, 0.3)])
def test_integrators_and_hybrid_flux(oracle, scheme, ctu, thr):
    flux = "HLLE/HLLC" if thr else "HLLC"
    P = config.Problem(_kh("minmod", flux, scheme, ctu, thr), fp_mode=abi.FP_STRICT)
    ref = oracle.OracleSim(P)
    ref.run(5)
    m = _gpu(P)
    m.run(5)
    assert np.array_equal(m.download(0), ref.get_state(0))
    m.close()


def test_field_reconstruction_options(oracle):
    for recon in ("minmod", "vanLeer", "O6", "WENO", "MC"):
        cfg = problems.orszag_tang_two_fluid_2d(24)
        cfg["states"]["field"]["reconstruction"] = recon
        P = config.Problem(cfg, fp_mode=abi.FP_STRICT)
        ref = oracle.OracleSim(P)
        ref.run(4)
        m = _gpu(P)
        m.run(4)
        for s in range(3):
            assert np.array_equal(m.download(s), ref.get_state(s)), (recon, s)
        m.close()


def test_fused_path_is_taken_for_2d_strang():
    for case in (1, 2, 3):
        name, cfg, _ = small_cases()[case]
        m = _gpu(config.Problem(cfg))
        m.run(1)
        assert m.last_advance_fused(), name
        m.close()
    name, cfg, _ = small_cases()[0]
    m = _gpu(config.Problem(cfg))
    m.run(1)
    assert m.last_advance_path() == 2      # 1-D and 3-D take the gather kernels
    m.close()
    name, cfg, _ = small_cases()[4]
    m = _gpu(config.Problem(cfg))
    m.run(1)
    assert m.last_advance_path() == 2
    m.close()


def test_unphysical_state_is_reported_not_aborted():
    """rho, p, T <= 0: the reference aborts (MFP_hydro_gas.cpp:152-164); the ABI returns CMFP_ERR_PHYSICS."""
    P = config.Problem(problems.kelvin_helmholtz_2d(32))
    m = _gpu(P)
    U = P.initial_data(0)
    U[4, 0, 10, 10] = -1.0      # negative total energy -> negative pressure
    m.upload(0, U)
    m.advance(1e-4)
    with pytest.raises(lib.CmfpError) as e:
        m.download(0)
    assert e.value.code == abi.ERR_PHYSICS
    # error flag + first offending cell.  The scan runs when the failure is detected, after the step
    # has spread the invalid value over the stencil reach of cell (10, 10)
    st, cell = m.last_error_cell()
    assert st == 0 and cell[2] == 0 and abs(cell[0] - 10) <= 4 and abs(cell[1] - 10) <= 4, (st, cell)
    assert "first offending cell" in str(e.value)
    # the flag is per failure: a valid state afterwards runs again
    m.upload(0, P.initial_data(0))
    m.advance(1e-4)
    m.download(0)
    m.close()


def test_ragged_sizes(oracle):
    """block sizes that are not multiples of the fused tile (14 x 14 cells)"""
    for n in ([17, 31], [45, 16]):
        cfg = problems.orszag_tang_two_fluid_2d(16)
        cfg["n_cell"] = n
        P = config.Problem(cfg, fp_mode=abi.FP_STRICT)
        ref = oracle.OracleSim(P)
        ref.run(3)
        for fused in (True, 2):
            m = _gpu(P, fused=fused)
            m.run(3)
            assert m.last_advance_path() == int(fused)
            for s in range(3):
                assert np.array_equal(m.download(s), ref.get_state(s)), (n, s, fused)
            m.close()


@pytest.mark.parametrize("flux,thr", [("HLLC", None), ("HLLE/HLLC", 0.3)])
def test_marching_kernel_strips_segments_and_walls(oracle, flux, thr):
    """The wave-marching hydro kernel (mfp_march.hip): blocks wider than one 60-column strip and
    taller than one segment (CMFP_MARCH_SEG is read when the library first launches it; the
    default segment is >= 32 rows), sizes that are no multiple of either, slip walls on all four
    domain faces (update_face_prim on x and y faces), single and split launches: bit-identical to
    the oracle and to the tiled kernel."""
    for n in ([61, 33], [125, 70], [180, 37], [250, 131]):
        cfg = _kh("vanLeer", flux, thr=thr, n=32)
        cfg["n_cell"] = n
        cfg["is_periodic"] = [0, 0]
        wall = {"lo": {"fill_hydro_bc": "slipwall"}, "hi": {"fill_hydro_bc": "slipwall"}}
        cfg["states"]["fluid"]["bc"] = {"x": wall, "y": wall}
        P = config.Problem(cfg, fp_mode=abi.FP_STRICT)
        ref = oracle.OracleSim(P)
        ref.run(4)
        for fused, overlap in ((True, 1), (True, 2), (3, 1)):
            m = _gpu(P, fused=fused, overlap=overlap)
            m.run(4)
            assert m.last_advance_path() == 1
            assert np.array_equal(m.download(0), ref.get_state(0)), (n, fused, overlap)
            m.close()


def test_full_size_properties():
    """At BASELINE size (2048^2 KH) the oracle is too slow; check what the scheme
    guarantees: exact periodic conservation to rounding, fused == staged kernels
    bit for bit (STRICT mode; FMA contraction differs between the two in FAST
    mode, checked to 1e-12 below), positivity."""
    P = config.Problem(problems.kelvin_helmholtz_2d(2048), fp_mode=abi.FP_STRICT)
    a, b = _gpu(P, fused=True), _gpu(P, fused=False)
    U0 = P.initial_data(0)
    a.run(3)
    b.run(3)
    A, B = a.download(0), b.download(0)
    assert np.array_equal(A, B)
    for c in range(5):
        tot0, tot1 = U0[c].sum(dtype=np.longdouble), A[c].sum(dtype=np.longdouble)
        assert abs(tot1 - tot0) <= 1e-11 * np.abs(U0[c]).sum(dtype=np.longdouble)
    assert A[0].min() > 0
    a.close()
    b.close()
    Pf = config.Problem(problems.kelvin_helmholtz_2d(2048), fp_mode=abi.FP_FAST)
    f = _gpu(Pf, fused=True)
    f.run(3)
    assert rel_linf(f.download(0), A) <= 1e-12
    f.close()


# ---- collisions_rambo -------------------------------------------------------------------
@pytest.mark.parametrize("neutral", [False, True])
def test_collisions_relaxation_parity(oracle, neutral):
    """Collisions alone, RK2 (stage-by-stage kernels).  STRICT: the hard-sphere branch (sqrt) is
    bit-identical; the Coulomb branch calls pow(x, -1.5), where the device library and glibc may
    differ in the last place, hence 1e-13."""
    for mode, tol in ((abi.FP_STRICT, 0.0 if neutral else 1e-13), (abi.FP_FAST, 1e-12)):
        P = config.Problem(problems.collisions_relaxation(8, neutral), fp_mode=mode)
        ref = oracle.OracleSim(P)
        m = _gpu(P)
        for _ in range(40):
            ref.advance(1e-4)
            m.advance(1e-4)
        for s in range(2):
            assert rel_linf(m.download(s), ref.get_state(s)) <= tol, (neutral, mode, s)
        m.close()


@pytest.mark.parametrize("passive", [False, True], ids=["mixture", "passive_labels"])
@pytest.mark.parametrize("neutral", [False, True])
def test_collisions_sub_component_parity(oracle, neutral, passive):
    """The species of the action is the second sub-component of a two-component state (gas.names + `species`,
    MFP_collisions.cpp:43-60): its share alpha of the density, its own mass and charge, the state's temperature;
    every sub-component's share accounted for and the tracer updated from the density share (:160-257).
    Parity unpinned (the reference holds no fixture for collisions): the oracle is the restatement."""
    for mode, tol in ((abi.FP_STRICT, 0.0 if neutral else 1e-13), (abi.FP_FAST, 1e-12)):
        P = config.Problem(problems.collisions_sub_component(8, neutral, passive), fp_mode=mode)
        ref = oracle.OracleSim(P)
        m = _gpu(P)
        for _ in range(40):
            ref.advance(1e-4)
            m.advance(1e-4)
        for s in range(2):
            assert rel_linf(m.download(s), ref.get_state(s)) <= tol, (neutral, passive, mode, s)
        if mode == abi.FP_STRICT and neutral:
            assert np.array_equal(m.download(0), ref.get_state(0))
        m.close()


@pytest.mark.parametrize("n", [(64, 64), (61, 37), (120, 9)])
def test_field_materials_select_the_kernel(oracle, n):
    """The marching vacuum kernel takes the 2-D O6 field only where mu = ep = 1 on the whole block (checked on the
    device after every upload); with a dielectric somewhere the tiled kernel runs.  Both bit-identical to the oracle
    (STRICT), on block sizes that are no multiple of the 56-column strips / row segments; an upload of vacuum data
    after dielectric data switches back."""
    def problem(dielectric):
        cfg = problems.orszag_tang_two_fluid_2d(64)
        cfg["n_cell"] = list(n)
        if dielectric:
            cfg["states"]["field"]["value"]["ep"] = lambda d: 1.0 + 0.25 * (np.sin(d["x"]) > 0.5)
            cfg["states"]["field"]["value"]["mu"] = lambda d: 1.0 + 0.5 * (np.cos(d["y"]) > 0.3)
        return config.Problem(cfg, fp_mode=abi.FP_STRICT)
    for dielectric in (True, False):
        P = problem(dielectric)
        ref = oracle.OracleSim(P)
        dts_ref = ref.run(5)
        m = _gpu(P)
        assert m.run(5) == dts_ref and m.last_advance_path() == 1
        for s in range(3):
            assert np.array_equal(m.download(s), ref.get_state(s)), (dielectric, P.state_names[s])
        if dielectric:      # the same context, re-initialised with vacuum data: the check is made again
            Pv = problem(False)
            for s in range(3):
                m.upload(s, Pv.initial_data(s))
            m.time, m.dt, m.step = 0.0, None, 0
            rv = oracle.OracleSim(Pv)
            assert m.run(5) == rv.run(5)
            for s in range(3):
                assert np.array_equal(m.download(s), rv.get_state(s)), ("re-upload", P.state_names[s])
        m.close()


def test_force_dt(oracle):
    """force_dt > 0 is what MFP::estTimeStep returns (MFP_timestep.cpp:8): per-step driver, device-resident
    dt loop and oracle all step with exactly force_dt; states bit-identical (hard-sphere branch)."""
    P = config.Problem(problems.collisions_relaxation(8, True), fp_mode=abi.FP_STRICT)
    ref = oracle.OracleSim(P)
    assert ref.run(6) == [1e-4] * 6
    m = _gpu(P)
    assert m.est_time_step() == 1e-4
    assert m.run(6) == [1e-4] * 6
    for s in range(2):
        assert np.array_equal(m.download(s), ref.get_state(s))
    m.close()
    # Strang two-fluid problem on the fused kernels: the device-side computeNewDt keeps dt = force_dt
    cfg = problems.orszag_tang_two_fluid_2d(32)
    cfg["force_dt"] = 2.5e-4
    P = config.Problem(cfg, fp_mode=abi.FP_STRICT)
    ref = oracle.OracleSim(P)
    assert ref.run(5) == [2.5e-4] * 5
    m = _gpu(P, fused=True)
    assert m.run(5) == [2.5e-4] * 5 and m.dt == 2.5e-4 and m.last_advance_fused()
    for s in range(3):
        assert np.array_equal(m.download(s), ref.get_state(s))
    m.close()


def test_collisions_with_plasma5_fused(oracle):
    """two-fluid Orszag-Tang with ion-electron collisions beside plasma5: the fused Strang step
    evaluates both sources in one kernel; the staged path runs them as two actions."""
    cfg = problems.orszag_tang_two_fluid_2d(32, collisions=True)
    for mode, tol in ((abi.FP_STRICT, 1e-13), (abi.FP_FAST, 1e-12)):
        P = config.Problem(cfg, fp_mode=mode)
        ref = oracle.OracleSim(P)
        ref.run(6)
        refc = oracle.OracleSim(config.Problem(problems.orszag_tang_two_fluid_2d(32), fp_mode=mode))
        refc.run(6)
        assert rel_linf(ref.get_state(0), refc.get_state(0)) > 1e-6      # the collisions do something here
        for fused in (False, True):
            m = _gpu(P, fused=fused)
            m.run(6)
            assert m.last_advance_fused() == fused
            for s in range(3):
                assert rel_linf(m.download(s), ref.get_state(s)) <= tol, (mode, fused, P.state_names[s])
            m.close()


def test_full_size_two_fluid_properties():
    """two-fluid Orszag-Tang at 1024^2 (the oracle is too slow here): the tiled kernels, the gather
    kernels and the stage-by-stage kernels agree bit for bit in STRICT mode, species mass is
    conserved to rounding on the periodic domain, mu and ep stay exactly 1."""
    P = config.Problem(problems.orszag_tang_two_fluid_2d(1024), fp_mode=abi.FP_STRICT)
    runs = []
    for fused in (True, 2, False):
        m = _gpu(P, fused=fused)
        dts = m.run(3)
        assert m.last_advance_path() == int(fused)
        runs.append((dts, [m.download(s) for s in range(3)]))
        m.close()
    for dts, data in runs[1:]:
        assert dts == runs[0][0]
        for s in range(3):
            assert np.array_equal(data[s], runs[0][1][s]), s
    # interior / frame split launches on two streams (what a rank of a decomposed run does), source epilogue
    # included: same bits (the frame launch of the field must wait for the species' interior launches)
    m = _gpu(P, overlap=2)
    dts = m.run(3)
    assert dts == runs[0][0]
    for s in range(3):
        assert np.array_equal(m.download(s), runs[0][1][s]), ("split launches", s)
    m.close()
    data = runs[0][1]
    for s, name in enumerate(P.state_names):
        if name == "field":
            assert np.all(data[s][8] == 1.0) and np.all(data[s][9] == 1.0)
        else:
            m0 = P.initial_data(s)[0].sum(dtype=np.longdouble)
            assert abs(data[s][0].sum(dtype=np.longdouble) - m0) <= 1e-12 * abs(m0)


def test_full_size_3d_two_fluid_properties():
    """two-fluid Richtmyer-Meshkov at 96 x 100 x 112 (the oracle is too slow here): the z-marching hydro /
    plane-tile field kernels with the source epilogue, the gather kernels and the stage-by-stage
    kernels agree bit for bit in STRICT mode (states and dt sequence); FAST stays within 1e-12."""
    P = config.Problem(problems.richtmyer_meshkov_3d(96, 100, 112), fp_mode=abi.FP_STRICT)
    runs = []
    for fused in (True, 2, False):
        m = _gpu(P, fused=fused)
        dts = m.run(3)
        assert m.last_advance_path() == (2 if fused else 0)
        runs.append((dts, [m.download(s) for s in range(3)]))
        m.close()
    for dts, data in runs[1:]:
        assert dts == runs[0][0]
        for s in range(3):
            assert np.array_equal(data[s], runs[0][1][s]), s
    m = _gpu(config.Problem(problems.richtmyer_meshkov_3d(96, 100, 112), fp_mode=abi.FP_FAST))
    dts = m.run(3)
    np.testing.assert_allclose(dts, runs[0][0], rtol=1e-12)
    for s in range(3):
        a, b = m.download(s), runs[0][1][s]
        assert rel_linf(a, b) <= 1e-12, (s, rel_linf(a, b))
    m.close()


# ---- viscous hydro fluxes -----------------------------------------------------------------
@pytest.mark.parametrize("visc", [{"type": "PowerLaw", "mu0": 0.08, "T0": 1 / 1.38064852e-23, "n": 1, "Pr": 0.71},
                                  {"type": "Sutherland", "mu0": 0.05, "T0": 1 / 1.38064852e-23, "S": 0.4 / 1.38064852e-23, "Pr": 0.71},
                                  {"type": "UserDefined", "mu0": 0.03, "Pr": 0.9}])
@pytest.mark.parametrize("recon,scheme", [("O6", "euler"), ("minmod", "strang")])
def test_viscous_fluxes_parity(oracle, visc, recon, scheme):
    """HydroState::calc_viscous_fluxes + the viscous dt limit, all three viscosity laws, STRICT:
    bit-identical to the oracle (states and dt sequence)."""
    cfg = problems.viscous_vortex_2d(48, recon=recon)
    cfg["time_integration_scheme"] = scheme
    cfg["states"]["air"]["viscosity"] = visc
    P = config.Problem(cfg, fp_mode=abi.FP_STRICT)
    ref = oracle.OracleSim(P)
    dts_ref = ref.run(8)
    m = _gpu(P)
    dts = m.run(8)
    assert m.last_advance_path() == 0          # viscous states run on the stage-by-stage kernels
    assert dts == dts_ref
    assert np.array_equal(m.download(0), ref.get_state(0))
    inviscid = problems.viscous_vortex_2d(48, recon=recon)
    del inviscid["states"]["air"]["viscosity"]
    inviscid["time_integration_scheme"] = scheme
    r0 = oracle.OracleSim(config.Problem(inviscid, fp_mode=abi.FP_STRICT))
    r0.run(8)
    assert rel_linf(r0.get_state(0), ref.get_state(0)) > 1e-6      # the viscous terms matter in this test
    m.close()


def test_reference_viscous_vortex_acceptance():
    """The reference's own acceptance test for this path (Exec/testing/Viscous-Vortex/check.py):
    run the decaying Lamb-Oseen vortex to t = 20 and require the speed within 1 % of the analytic
    v_theta(r, t + t0) for r <= 8.  The reference resolves r <= 8 with dx = 40/384 through two AMR
    levels; here the whole domain has that resolution."""
    P = config.Problem(problems.viscous_vortex_2d(384), fp_mode=abi.FP_FAST)
    m = _gpu(P)
    stop = 20.0
    dt = m.est_time_step()
    while m.time < stop - 1e-12:
        dt = min(dt, stop - m.time)
        m.advance(dt)
        dt = m.L.cmfp_compute_new_dt(m.est_time_step(), dt)
    U = m.download(0)
    m.close()
    c = P.cell_centres()
    x, y = c["x"][0], c["y"][0]
    u, v = U[1, 0] / U[0, 0], U[2, 0] / U[0, 0]
    R = np.sqrt(x * x + y * y)
    vel = np.sqrt(u * u + v * v)
    nu, G, t0 = 0.08, 1.0, 10.0
    vt = G / (2 * np.pi * R) * (1 - np.exp(-R ** 2 / (4 * nu * (stop + t0))))
    sel = R <= 8.0
    rel = np.abs((vel[sel] - vt[sel]) / vt[sel])
    assert rel.max() <= 0.01, rel.max()


# ---- passive scalars (tracers) ----------------------------------------------------------------
def _band(d):
    return (np.abs(d["y"] - 0.5) < 0.25).astype(float)


@pytest.mark.parametrize("flux,scheme,recon", [("HLLC", "strang", "vanLeer"), ("HLLE", "strang", "minmod"),
                                                ("HLLC", "RK2", "MC"), ("HLLC", "euler", "O6")])
def test_tracers_parity_2d(oracle, flux, scheme, recon):
    """hydro state with two passive scalars: STRICT bit-identical to the oracle, FAST within 1e-12, walls in y (scalar
    ghost rule on the tracer comps).  Strang steps with a three-point limiter run on the marching kernel's tracer
    instances (k_ctu_march<..., NT>), and once more on the stage-by-stage kernels; euler / RK2 / O6 stage by stage."""
    for walls in (False, True):
        cfg = _kh(recon, flux, scheme)
        if walls:
            cfg["is_periodic"] = [1, 0]
            wall = {"lo": {"fill_hydro_bc": "slipwall"}, "hi": {"fill_hydro_bc": "slipwall"}}
            cfg["states"]["fluid"]["bc"] = {"y": wall}
        problems.with_tracers(cfg, "fluid", [_band, lambda d: 0.25 + 0.5 * d["x"]])
        for mode, tol in ((abi.FP_STRICT, 0.0), (abi.FP_FAST, 1e-12)):
            P = config.Problem(cfg, fp_mode=mode)
            ref = oracle.OracleSim(P)
            dts_ref = ref.run(5)
            for fused in ((True, False) if scheme == "strang" else (True,)):
                m = _gpu(P, fused=fused)
                dts = m.run(5)
                assert m.last_advance_path() == (1 if scheme == "strang" and fused else 0)
                if tol == 0.0:
                    assert dts == dts_ref
                assert rel_linf(m.download(0), ref.get_state(0)) <= tol, (walls, mode, fused)
                for c in (5, 6):
                    assert rel_linf(m.download(0)[c], ref.get_state(0)[c]) <= tol, (walls, mode, c, fused)
                if tol == 0.0:
                    assert np.array_equal(m.download(0), ref.get_state(0)), (walls, fused)
                m.close()


@pytest.mark.parametrize("n,nt", [((150, 70), 2), ((64, 131), 1), ((61, 9), 1)])
def test_tracers_on_the_marching_kernel_at_size(oracle, n, nt):
    """several strips and segments, block sizes off the 60-column strips, one and two passive scalars, outflow faces
    in x and periodic y (the in-kernel wrap), two species in one launch with the plasma5 source beside them: STRICT
    bit-identical to the oracle, FAST within 1e-12, path 1."""
    cfg = problems.orszag_tang_two_fluid_2d(64)
    cfg["n_cell"] = list(n)
    cfg["is_periodic"] = [0, 1]
    al = [lambda d: 0.5 + 0.25 * np.sin(d["x"]) * np.cos(2 * d["y"]), lambda d: 0.1 + 0.05 * np.cos(d["x"])][:nt]
    for name in ("ion", "electron"):
        problems.with_tracers(cfg, name, al)
    for mode, tol in ((abi.FP_STRICT, 0.0), (abi.FP_FAST, 1e-12)):
        P = config.Problem(cfg, fp_mode=mode)
        ref = oracle.OracleSim(P)
        dts_ref = ref.run(4)
        m = _gpu(P)
        dts = m.run(4)
        assert m.last_advance_path() == 1
        assert m.last_source_fusion() == 3      # stage A as the prologue of the tracer instances too, stage C the field's epilogue
        for s in range(3):
            a, b = m.download(s), ref.get_state(s)
            if tol == 0.0:
                assert dts == dts_ref and np.array_equal(a, b), (n, nt, P.state_names[s])
            else:
                assert rel_linf(a, b) <= tol, (n, nt, P.state_names[s])
        m.close()


def test_tracers_two_fluid_and_3d(oracle):
    """tracers on the ion state of the two-fluid Orszag-Tang problem (plasma5 source beside them)
    and on a 3-D two-fluid block: STRICT bit-identical to the oracle."""
    cfg = problems.orszag_tang_two_fluid_2d(24)
    problems.with_tracers(cfg, "ion", [lambda d: 0.5 + 0.25 * np.sin(d["x"])])
    cfg3 = problems.richtmyer_meshkov_3d(16, 8, 8)
    problems.with_tracers(cfg3, "electron", [lambda d: (d["x"] > 0).astype(float), 0.5])
    for c in (cfg, cfg3):
        P = config.Problem(c, fp_mode=abi.FP_STRICT)
        ref = oracle.OracleSim(P)
        ref.run(4)
        m = _gpu(P)
        m.run(4)
        for s in range(m.n_states):
            assert np.array_equal(m.download(s), ref.get_state(s)), (P.dim, P.state_names[s])
        m.close()


def test_tracers_with_hybrid_flux_are_refused():
    cfg = _kh("minmod", "HLLE/HLLC", thr=0.3)
    problems.with_tracers(cfg, "fluid", [0.5])
    with pytest.raises(lib.CmfpError) as e:
        _gpu(config.Problem(cfg))
    assert e.value.code == abi.ERR_INVALID


# ---- Dirichlet (inflow) faces: bc.<dim>.<side>.<primitive> values and functions ------------------
def _inflow(scheme="strang", tracers=False, time_dep=False):
    from test_oracle import inflow_cfg
    cfg = inflow_cfg(32, 12, scheme)
    lo = cfg["states"]["fluid"]["bc"]["x"]["lo"]
    if time_dep:
        f = lambda d: 3.0 + 0.5 * np.sin(40.0 * d["t"]) + 0.0 * d["x"]
        f.uses_time = True
        lo["x_vel"] = f
    if tracers:
        problems.with_tracers(cfg, "fluid", [0.25, lambda d: d["y"]])
        lo["alpha"] = [lambda d: (d["y"] > 0.2).astype(float), 0.5]
    return cfg


@pytest.mark.parametrize("scheme,tracers,time_dep", [("strang", False, False), ("RK2", False, False), ("euler", True, False),
                                                     ("strang", True, True), ("strang", False, True)])
def test_dirichlet_faces_parity(oracle, scheme, tracers, time_dep):
    """inflow face with a profile, a wall face with a prescribed tangential velocity, optionally
    tracers and a time-dependent inflow velocity: STRICT bit-identical to the oracle, FAST 1e-12."""
    for mode, tol in ((abi.FP_STRICT, 0.0), (abi.FP_FAST, 1e-12)):
        P = config.Problem(_inflow(scheme, tracers, time_dep), fp_mode=mode)
        ref = oracle.OracleSim(P)
        dts_ref = ref.run(8)
        m = _gpu(P)
        dts = m.run(8)
        # Strang steps of a state without tracers: the marching kernel with the face tables (mfp_march.hip, TAB);
        # euler / RK2 and tracer states: stage-by-stage kernels
        assert m.last_advance_path() == (1 if scheme == "strang" and not tracers else 0)
        if tol == 0.0:
            assert dts == dts_ref
        assert rel_linf(m.download(0), ref.get_state(0)) <= tol, (scheme, mode)
        m.close()
        if scheme == "strang" and not tracers:      # ... and the same through the staged kernels and with split launches
            for kw in ({"fused": False}, {"overlap": 2}):
                m2 = _gpu(P, **kw)
                m2.run(8)
                assert m2.last_advance_path() == (0 if "fused" in kw else 1)
                assert rel_linf(m2.download(0), ref.get_state(0)) <= tol, (scheme, mode, kw)
                m2.close()


def test_dirichlet_field_face_and_1d(oracle):
    """a driven field face (transverse B prescribed as a function of time at x-lo of the 1-D
    two-fluid problem: the Waveguide-style source) beside hydro outflow faces"""
    cfg = problems.brio_wu_two_fluid_1d(96)
    f = lambda d: 1.0 + 0.1 * np.sin(200.0 * d["t"]) + 0.0 * d["x"]
    f.uses_time = True
    cfg["states"]["field"]["bc"] = {"x": {"lo": {"y_B": f, "z_D": 0.0}, "hi": {}}}
    P = config.Problem(cfg, fp_mode=abi.FP_STRICT)
    ref = oracle.OracleSim(P)
    dts_ref = ref.run(10)
    m = _gpu(P)
    dts = m.run(10)
    assert dts == dts_ref
    for s in range(3):
        assert np.array_equal(m.download(s), ref.get_state(s)), P.state_names[s]
    base = oracle.OracleSim(config.Problem(problems.brio_wu_two_fluid_1d(96), fp_mode=abi.FP_STRICT))
    base.run(10)
    fs = P.state_index["field"]
    assert np.abs(base.get_state(fs) - ref.get_state(fs)).max() > 1e-6     # the driven face does something
    m.close()


# ---- z-marching 3-D hydro kernel (mfp_march3d.hip) ------------------------------------------------
def _hydro3d(recon, flux, n, walls):
    cfg = problems.richtmyer_meshkov_3d(n[0], n[1], n[2])
    for k in ("electron", "field"):
        del cfg["states"][k]
    cfg["actions"] = {"fluxes": {"type": "CTU", "corner_transport": True, "states": ["ion"]}}
    cfg["states"]["ion"].update({"reconstruction": recon, "flux": flux})
    if walls:
        cfg["is_periodic"] = [0, 0, 0]
        wall = {"lo": {"fill_hydro_bc": "slipwall"}, "hi": {"fill_hydro_bc": "slipwall"}}
        cfg["states"]["ion"]["bc"] = {"x": wall, "y": wall, "z": {"lo": {"fill_hydro_bc": "outflow"}, "hi": wall["hi"]}}
    return cfg


@pytest.mark.parametrize("recon,flux,walls", [("minmod", "HLLC", False), ("vanLeer", "HLLE", True), ("MC", "HLLC", True),
                                              ("centre", "HLLC", False)])
def test_march3d_hydro(oracle, recon, flux, walls):
    """hydro-only 3-D block on the z-marching kernel: partial tiles in x and y, three z-segments
    (40 planes, 16 per segment), walls / outflow on all six faces, fused CFL reduction: bit-identical
    to the oracle and to the gather kernels."""
    P = config.Problem(_hydro3d(recon, flux, [20, 17, 40], walls), fp_mode=abi.FP_STRICT)
    ref = oracle.OracleSim(P)
    dts_ref = ref.run(3)
    for fused in (True, 2):
        m = _gpu(P, fused=fused)
        dts = m.run(3)
        assert m.last_advance_path() == (1 if fused is True else 2)
        assert dts == dts_ref, (fused, dts, dts_ref)
        assert np.array_equal(m.download(0), ref.get_state(0)), (recon, flux, walls, fused)
        m.close()


# ---- 3-D field update from LDS plane tiles (mfp_field3d.hip) ---------------------------------------
@pytest.mark.parametrize("recon,walls,n", [("O6", False, (20, 17, 40)), ("O6", True, (20, 17, 40)), ("vanLeer", True, (20, 17, 40)),
                                           ("minmod", True, (20, 17, 40)), ("O6", True, (28, 14, 30)), ("O6", False, (29, 15, 33)),
                                           ("O6", True, (6, 4, 9)), ("O6", False, (16, 32, 14))])
def test_field3d_tiles(oracle, recon, walls, n):
    """two-fluid + Maxwell 3-D block on the plane-tile field kernels: partial tiles in every direction,
    several marching segments, conducting walls / outflow on all six faces of the field: bit-identical
    to the oracle and to the gather kernels."""
    cfg = problems.richtmyer_meshkov_3d(*n)     # tile edges: multiples of 14 / 16, one more, and blocks below one tile
    cfg["states"]["field"]["reconstruction"] = recon
    if walls:
        cfg["is_periodic"] = [0, 0, 0]
        hw = {"lo": {"fill_hydro_bc": "slipwall"}, "hi": {"fill_hydro_bc": "slipwall"}}
        fw = {k: {"fill_D_bc": "asymmetry", "fill_B_bc": "asymmetry", "fill_phi_bc": "symmetry", "fill_psi_bc": "symmetry",
                  "fill_mu_bc": "symmetry", "fill_ep_bc": "symmetry"} for k in ("lo", "hi")}
        fo = {"lo": {}, "hi": fw["hi"]}
        for name in ("ion", "electron"):
            cfg["states"][name]["bc"] = {"x": hw, "y": hw, "z": hw}
        cfg["states"]["field"]["bc"] = {"x": fw, "y": fo, "z": fw}
    P = config.Problem(cfg, fp_mode=abi.FP_STRICT)
    ref = oracle.OracleSim(P)
    dts_ref = ref.run(3)
    for fused in (True, 2):
        m = _gpu(P, fused=fused)
        dts = m.run(3)
        assert dts == dts_ref, (fused, dts, dts_ref)
        for s in range(3):
            assert np.array_equal(m.download(s), ref.get_state(s)), (recon, walls, fused, P.state_names[s])
        m.close()


# ---- Strang stage C as the epilogue of the tiled field kernel (mfp_fused.hip) ----------------------
@pytest.mark.parametrize("name,make", [
    ("ot_plasma5", lambda: problems.orszag_tang_two_fluid_2d(48)),
    ("ot_lorentz", lambda: problems.orszag_tang_two_fluid_2d(48, source="Lorentz")),
    ("gem_walls", lambda: problems.gem_reconnection_2d(48, 34)),
    ("rmi_3d", lambda: problems.richtmyer_meshkov_3d(20, 17, 40)),
])
def test_source_epilogue(oracle, monkeypatch, name, make):
    """the trailing source half step evaluated inside the field's CTU kernel (one source launch per
    step instead of two) leaves the same bits as the separate launch and as the oracle, dt included"""
    P = config.Problem(make(), fp_mode=abi.FP_STRICT)
    ref = oracle.OracleSim(P)
    dts_ref = ref.run(4)
    for epilogue in (True, False):
        monkeypatch.setenv("CMFP_SRC_EPILOGUE", "1" if epilogue else "0")
        m = _gpu(P)
        m.profile_enable(True)
        m.profile_read(2)
        dts = m.run(4)
        launches, _ = m.profile_read(2)
        # with the epilogue the 2-D steps also take stage A as the prologue of the marching kernels (STRICT since
        # round 3): no source launch at all; 3-D keeps stage A a launch
        assert launches == ((0 if P.dim == 2 else 4) if epilogue else 8), (name, epilogue, launches)
        assert dts == dts_ref, (name, epilogue)
        for s in range(len(P.state_names)):
            assert np.array_equal(m.download(s), ref.get_state(s)), (name, epilogue, P.state_names[s])
        m.close()


def test_source_epilogue_with_split_launches_and_mixed_stencils(oracle):
    """forced split launches (overlap=2): with three-point species and an O6 field the field's interior
    launch carries the source epilogue on the second stream (its interior lies inside the species'
    interiors); with O6 species the hydro states leave the tiled kernels, nothing is split and the epilogue
    follows whole launches.  Bit-identical to the oracle either way, at most one source launch per step."""
    for hydro_recon, field_recon in (("O6", "minmod"), ("minmod", "O6"), ("vanLeer", "minmod")):
        cfg = problems.orszag_tang_two_fluid_2d(96)
        for name in ("ion", "electron"):
            cfg["states"][name]["reconstruction"] = hydro_recon
        cfg["states"]["field"]["reconstruction"] = field_recon
        P = config.Problem(cfg, fp_mode=abi.FP_STRICT)
        ref = oracle.OracleSim(P)
        dts_ref = ref.run(3)
        m = _gpu(P, overlap=2)
        m.profile_enable(True)
        m.profile_read(2)
        dts = m.run(3)
        launches, _ = m.profile_read(2)
        # (minmod species with the O6 vacuum field: the marching kernels, which since round 3 also take stage A as
        # their prologue in STRICT arithmetic: no source launch at all)
        assert launches == (0 if (hydro_recon, field_recon) == ("minmod", "O6") else 3), (hydro_recon, field_recon, launches)
        assert dts == dts_ref
        for s in range(3):
            assert np.array_equal(m.download(s), ref.get_state(s)), (hydro_recon, field_recon, P.state_names[s])
        m.close()
