"""Gases of several species with different mass / gamma (ThermallyPerfectGas::get_gamma_from_prim / _from_cons,
get_cp_*, MFP_thermally_perfect_gas.cpp:51-177; HydroGas::get_mass_from_*, MFP_hydro_gas.cpp:50-96): gamma, mass
and cp of a cell follow its mass fractions alpha_i = tracer_i / rho.  No shipped example of the reference uses such
a gas (Thermal-RMI's `alpha` values sit on a scalar `mass`), so nothing the reference holds pins this branch:
PARITY UNPINNED beyond the restatement; the acoustic check below pins the mixture rule itself (a sound wave in a
uniform mixture travels at sqrt(gamma_mix p / rho), gamma_mix = sum(alpha cp) / sum(alpha cv))."""
import math

import numpy as np
import pytest

from cerberus_amd import abi, config, lib, problems
from helpers import rel_linf


def _acoustic_cfg(n, alpha, masses, gammas, recon="vanLeer", eps=1e-5):
    cfg = problems.kelvin_helmholtz_2d(8, recon=recon)
    cfg["n_cell"] = [n, 4]
    cfg["prob_hi"] = [1.0, 4.0 / n]
    k = 2 * math.pi
    cfg["states"]["fluid"]["value"] = {"rho": 1.0, "x_vel": lambda d: eps * np.sin(k * d["x"]), "y_vel": 0.0,
                                       "z_vel": 0.0, "p": 1.0}
    return problems.with_mixture(cfg, "fluid", alpha, masses, gammas)


ACOUSTIC_TOL = 1e-2      # velocity amplitude of the standing wave over one period against cos(w t): 1 % (of the amplitude)


def check_acoustic(make, alpha, masses, gammas, n=128):
    """u = eps sin(k x) on a uniform mixture at rest: a standing sound wave, u(t) = eps sin(k x) cos(a k t)."""
    eps = 1e-5
    P = config.Problem(_acoustic_cfg(n, alpha, masses, gammas, eps=eps))
    run = make(P)
    al = list(alpha) + [1.0 - sum(alpha)]
    cp = sum(a * g / (m * (g - 1)) for a, m, g in zip(al, masses, gammas))
    cv = sum(a * 1 / (m * (g - 1)) for a, m, g in zip(al, masses, gammas))
    a0 = math.sqrt(cp / cv * 1.0 / 1.0)
    k = 2 * math.pi
    w, T = a0 * k, 1.0 / a0
    x = (np.arange(n) + 0.5) / n
    sn = np.sin(k * x)[None, :]
    nsteps = int(math.ceil(T / run.est()))
    dt, t, err = T / nsteps, 0.0, 0.0
    for _ in range(nsteps):
        run.advance(dt)
        t += dt
        U = run.state(0)
        amp = 2.0 * float(((U[1, 0] / U[0, 0]) * sn).mean()) / eps
        err = max(err, abs(amp - math.cos(w * t)))
    U = run.state(0)
    drift = float(np.abs(U[5] / U[0] - alpha[0]).max())      # the composition stays uniform
    run.close()
    return err, drift


MIXES = [([0.3], [1.0, 0.138], [1.4, 5.0 / 3.0]),           # air / helium
         ([0.2, 0.5], [1.0, 2.0, 0.5], [1.4, 1.1, 5.0 / 3.0])]


@pytest.mark.parametrize("alpha,masses,gammas", MIXES)
def test_mixture_sound_speed_oracle(oracle, alpha, masses, gammas):
    from test_physics_pins import OracleRunner
    err, drift = check_acoustic(lambda P: OracleRunner(oracle, P), alpha, masses, gammas)
    assert err <= ACOUSTIC_TOL and drift <= 1e-12, (err, drift)
    # the same wave with the gamma of species 0 alone would be off by far more than the tolerance
    al = list(alpha) + [1.0 - sum(alpha)]
    cp = sum(a * g / (m * (g - 1)) for a, m, g in zip(al, masses, gammas))
    cv = sum(a / (m * (g - 1)) for a, m, g in zip(al, masses, gammas))
    assert abs(math.sqrt(cp / cv) - math.sqrt(gammas[0])) * 2 * math.pi / math.sqrt(cp / cv) > 10 * ACOUSTIC_TOL


def test_mixture_of_equal_species_is_the_tracer_case(oracle):
    """mass_const and gamma_const (MFP_thermally_perfect_gas.cpp:32-34): the constants of species 0, bit for bit."""
    a = problems.shock_bubble_2d(32, 16)
    b = problems.shock_bubble_2d(32, 16)
    problems.with_mixture(a, "fluid", a["states"]["fluid"]["value"]["alpha"], [1.0, 1.0], [1.4, 1.4])
    b["states"]["fluid"]["gas"].update(mass=[1.0, 1.0], gamma=[1.4, 1.4])
    Pa, Pb = config.Problem(a), config.Problem(b)
    assert Pa.state_descs()[0].mixture == 0
    ra, rb = oracle.OracleSim(Pa), oracle.OracleSim(Pb)
    assert ra.run(3) == rb.run(3) and np.array_equal(ra.get_state(0), rb.get_state(0))


def test_mixture_descriptor_and_initial_data():
    P = config.Problem(problems.shock_bubble_2d(32, 16, species=3))
    d = P.state_descs()[0]
    assert d.mixture == 1 and d.n_tracers == 2 and P.n_comp(0) == 7
    assert list(d.species_mass) == [1.0, 2.0, 0.138] and d.mass == 1.0 and d.gamma == 1.4
    U = P.initial_data(0)
    al = [U[5] / U[0], U[6] / U[0]]
    g = config.gas_gamma_from_alpha(P.cfg["states"]["fluid"]["gas"], al)
    ke = 0.5 * (U[1] ** 2 + U[2] ** 2 + U[3] ** 2) / U[0]
    p = (U[4] - ke) * (g - 1)
    assert np.abs(p[:, :, 20:] - 1.0).max() < 1e-14      # pressure equilibrium across the bubble and the band
    assert g.min() < 1.3 and g.max() > 1.6


# ions of two kinds in one state (same charge, different mass; different charge and mass), electrons of two "kinds"
PLASMA_MIXES = [{"ion": ([0.4], [1.0, 1.5], [1.0, 1.0])},
                {"ion": ([0.5], [1.0, 4.0], [1.0, 2.0]), "electron": ([0.25], [0.04, 0.08], [-1.0, -1.0])}]


@pytest.mark.parametrize("mix", PLASMA_MIXES)
def test_mixture_plasma_frequency_oracle(oracle, mix):
    """the plasma5 source reads charge / mass of a cell from its mass fractions (MFP_plasma5.cpp:199-202): the
    Langmuir oscillation of a uniform plasma with mixture species runs at the plasma frequency of the mixture's r"""
    import test_physics_pins as pp
    err, back, uniform = pp.check_langmuir(lambda P: pp.OracleRunner(oracle, P), mix=mix)
    assert err <= pp.OSC_TOL and back <= pp.OSC_TOL and uniform <= 1e-12, (err, back, uniform)


def _viscous_mix_cfg(visc, recon, scheme, species):
    cfg = problems.viscous_vortex_2d(48, recon=recon)
    cfg["time_integration_scheme"] = scheme
    cfg["states"]["air"]["viscosity"] = visc
    blob = lambda d: 0.5 + 0.4 * np.tanh((np.sqrt(d["x"] ** 2 + d["y"] ** 2) - 4.0) / 1.5)
    if species == 0:       # equal species: passive scalars beside the viscous terms
        return problems.with_tracers(cfg, "air", [blob, 0.25])
    if species == 2:
        return problems.with_mixture(cfg, "air", [blob], [1.0, 0.4], [5 / 3, 1.4])
    return problems.with_mixture(cfg, "air", [lambda d: 0.8 * blob(d), 0.1], [1.0, 0.4, 2.0], [5 / 3, 1.4, 1.2])


VISC = [{"type": "PowerLaw", "mu0": 0.08, "T0": 1 / 1.38064852e-23, "n": 1, "Pr": 0.71},
        {"type": "Sutherland", "mu0": 0.05, "T0": 1 / 1.38064852e-23, "S": 0.4 / 1.38064852e-23, "Pr": 0.71},
        {"type": "UserDefined", "mu0": 0.03, "Pr": 0.9}]


def test_viscous_mixture_oracle_heat_flux_follows_cp(oracle):
    """kappa = mu cp / Pr with the cp of the mixture (get_cp_from_prim, MFP_hydro_viscous.cpp:147): the same run with
    the cp of species 0 alone differs, the passive-scalar case does not care"""
    a = oracle.OracleSim(config.Problem(_viscous_mix_cfg(VISC[0], "minmod", "strang", 2)))
    a.run(4)
    cfg = _viscous_mix_cfg(VISC[0], "minmod", "strang", 2)
    cfg["states"]["air"]["viscosity"] = dict(VISC[0], Pr=0.71 * 1.5)
    b = oracle.OracleSim(config.Problem(cfg))
    b.run(4)
    assert rel_linf(a.get_state(0), b.get_state(0)) > 1e-9 and np.isfinite(a.get_state(0)).all()


# ---- the HIP path against the oracle ---------------------------------------------------------------------------
def _gpu(P, **kw):
    from cerberus_amd.mfp import MFP
    m = MFP(P, **kw)
    m.init_data()
    return m


def _cases():
    out = []
    for recon, flux, scheme, species in (("vanLeer", "HLLC", "strang", 2), ("minmod", "HLLE", "strang", 3),
                                         ("MC", "HLLC", "RK2", 3), ("O6", "HLLC", "euler", 2), ("WENO", "HLLE", "strang", 2)):
        cfg = problems.shock_bubble_2d(48, 24, recon=recon, flux=flux, species=species)
        cfg["time_integration_scheme"] = scheme
        out.append((f"2d-{recon}-{flux}-{scheme}-{species}", cfg))
    cfg = problems.shock_bubble_2d(48, 24, species=3)
    cfg["is_periodic"] = [1, 1]
    del cfg["states"]["fluid"]["bc"]
    out.append(("2d-periodic", cfg))
    cfg = problems.shock_bubble_2d(48, 24)
    cfg["actions"]["fluxes"]["corner_transport"] = False
    out.append(("2d-no-corner-transport", cfg))
    out.append(("1d", problems.shock_bubble_2d(96, dim=1, species=3)))
    out.append(("3d", problems.shock_bubble_2d(32, 16, dim=3, nz=8, species=3)))
    # same gamma, different mass: only T (the validity test) and cp differ from the tracer case
    cfg = problems.shock_bubble_2d(48, 24)
    problems.with_mixture(cfg, "fluid", cfg["states"]["fluid"]["value"]["alpha"], [1.0, 0.138], [1.4, 1.4])
    out.append(("2d-mass-only", cfg))
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("name,cfg", _cases(), ids=[c[0] for c in _cases()])
def test_mixture_parity(oracle, name, cfg):
    """STRICT bit-identical to the oracle (data and dt sequence), FAST within 1e-12, on the stage-by-stage kernels."""
    for mode, tol in ((abi.FP_STRICT, 0.0), (abi.FP_FAST, 1e-12)):
        P = config.Problem(cfg, fp_mode=mode)
        ref = oracle.OracleSim(P)
        dts_ref = ref.run(6)
        m = _gpu(P)
        dts = m.run(6)
        assert m.last_advance_path() == 0
        got, want = m.download(0), ref.get_state(0)
        if tol == 0.0:
            assert dts == dts_ref, name
            assert np.array_equal(got, want), (name, rel_linf(got, want))
        else:
            assert rel_linf(got, want) <= tol, (name, mode)
        m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("scheme", ["strang", "euler"])
def test_mixture_inflow_parity(oracle, scheme):
    """a second gas blown in through a Dirichlet face (rho, velocity, p and alpha tabulated; gamma follows alpha) and a
    moving wall: STRICT bit-identical to the oracle, FAST 1e-12"""
    from test_oracle import inflow_cfg
    cfg = inflow_cfg(32, 12, scheme)
    problems.with_mixture(cfg, "fluid", [lambda d: 0.9 - 0.5 * d["y"]], [1.0, 0.25], [1.4, 5 / 3])
    cfg["states"]["fluid"]["bc"]["x"]["lo"]["alpha"] = [lambda d: (d["y"] > 0.2) * 0.8 + 0.1]
    for mode, tol in ((abi.FP_STRICT, 0.0), (abi.FP_FAST, 1e-12)):
        P = config.Problem(cfg, fp_mode=mode)
        ref = oracle.OracleSim(P)
        dts_ref = ref.run(8)
        m = _gpu(P)
        dts = m.run(8)
        got, want = m.download(0), ref.get_state(0)
        if tol == 0.0:
            assert dts == dts_ref and np.array_equal(got, want), rel_linf(got, want)
        else:
            assert rel_linf(got, want) <= tol
        m.close()
    # the inflowing gas arrives: the mass fraction next to the face follows the table, not the initial profile
    a = got[5, 0, :, 0] / got[0, 0, :, 0]
    y = (np.arange(12) + 0.5) * P.dx[1] + P.prob_lo[1]
    assert np.abs(a - ((y > 0.2) * 0.8 + 0.1)).max() < np.abs(a - (0.9 - 0.5 * y)).max()


@pytest.mark.gpu
@pytest.mark.parametrize("visc,recon,scheme,species", [(VISC[0], "O6", "euler", 2), (VISC[1], "minmod", "strang", 3),
                                                       (VISC[2], "vanLeer", "RK2", 2), (VISC[0], "minmod", "strang", 0)])
def test_viscous_mixture_parity(oracle, visc, recon, scheme, species):
    """viscous fluxes and the viscous dt limit of a mixture (T with the cell's mass, kappa with its cp, the
    stability limit with its gamma) and of a gas with passive scalars: STRICT bit-identical to the oracle"""
    P = config.Problem(_viscous_mix_cfg(visc, recon, scheme, species), fp_mode=abi.FP_STRICT)
    ref = oracle.OracleSim(P)
    dts_ref = ref.run(6)
    m = _gpu(P)
    dts = m.run(6)
    assert m.last_advance_path() == 0 and dts == dts_ref
    assert np.array_equal(m.download(0), ref.get_state(0))
    m.close()
    Pf = config.Problem(P.cfg, fp_mode=abi.FP_FAST)
    m = _gpu(Pf)
    m.run(6)
    assert rel_linf(m.download(0), ref.get_state(0)) <= 1e-12
    m.close()


@pytest.mark.gpu
def test_mixture_differs_from_single_gas(oracle):
    """the mixture rule is live on the HIP path: the same run with the constants of species 0 is far away"""
    cfg = problems.shock_bubble_2d(48, 24)
    m = _gpu(config.Problem(cfg))
    m.run(10)
    a = m.download(0)
    m.close()
    problems.with_mixture(cfg, "fluid", cfg["states"]["fluid"]["value"]["alpha"], [1.0, 1.0], [1.4, 1.4])
    m = _gpu(config.Problem(cfg))
    m.run(10)
    b = m.download(0)
    m.close()
    assert rel_linf(a[:5], b[:5]) > 1e-2


@pytest.mark.gpu
@pytest.mark.parametrize("alpha,masses,gammas", MIXES)
@pytest.mark.parametrize("mode", [abi.FP_STRICT, abi.FP_FAST])
def test_mixture_sound_speed_gpu(alpha, masses, gammas, mode):
    from test_physics_pins import GpuRunner
    err, drift = check_acoustic(lambda P: GpuRunner(config.Problem(P.cfg, fp_mode=mode)), alpha, masses, gammas)
    assert err <= ACOUSTIC_TOL and drift <= 1e-12, (err, drift)


def _plasma_cases():
    out = []
    wave = lambda d: 0.5 + 0.3 * np.sin(d["x"]) * np.cos(d["y"])
    for solver, source, scheme in (("explicit", "plasma5", "strang"), ("implicit", "plasma5", "strang"),
                                   ("explicit", "Lorentz", "strang"), ("explicit", "plasma5", "RK2")):
        cfg = problems.orszag_tang_two_fluid_2d(24, solver=solver, source=source)
        cfg["time_integration_scheme"] = scheme
        problems.with_mixture(cfg, "ion", [wave], [1.0, 3.0], [5 / 3, 1.4], [1.0, 2.0])
        if solver == "explicit" and scheme == "strang" and source == "plasma5":
            problems.with_mixture(cfg, "electron", [0.3, wave], [0.01, 0.02, 0.04], [5 / 3] * 3, [-1.0, -1.0, -2.0])
        out.append((f"ot-{source}-{solver}-{scheme}", cfg))
    cfg = problems.richtmyer_meshkov_3d(16, 8, 8)
    problems.with_mixture(cfg, "ion", [lambda d: (d["x"] > 0) * 0.8 + 0.1], [1.0, 2.0], [5 / 3, 5 / 3])
    out.append(("rmi3d", cfg))
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("name,cfg", _plasma_cases(), ids=[c[0] for c in _plasma_cases()])
def test_mixture_in_sources_parity(oracle, name, cfg):
    """mixture species in plasma5 (explicit, implicit) / Lorentz and the plasma dt limit: per-cell charge / mass.
    STRICT bit-identical to the oracle with the dt sequence (implicit: 1e-12, block elimination), FAST 1e-12."""
    for mode, tol in ((abi.FP_STRICT, 0.0), (abi.FP_FAST, 1e-12)):
        P = config.Problem(cfg, fp_mode=mode)
        ref = oracle.OracleSim(P)
        dts_ref = ref.run(5)
        m = _gpu(P)
        dts = m.run(5)
        assert m.last_advance_path() == 0
        exact = tol == 0.0 and "implicit" not in name
        if exact:
            assert dts == dts_ref, name
        for s in range(m.n_states):
            got, want = m.download(s), ref.get_state(s)
            if exact:
                assert np.array_equal(got, want), (name, s, rel_linf(got, want))
            else:
                assert rel_linf(got, want) <= max(tol, 1e-12), (name, mode, s)
        m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [abi.FP_STRICT, abi.FP_FAST])
@pytest.mark.parametrize("mix", PLASMA_MIXES)
def test_mixture_plasma_frequency_gpu(mix, mode):
    import test_physics_pins as pp
    err, back, uniform = pp.check_langmuir(lambda P: pp.GpuRunner(config.Problem(P.cfg, fp_mode=mode)), mix=mix)
    assert err <= pp.OSC_TOL and back <= pp.OSC_TOL and uniform <= 1e-12, (err, back, uniform)


def _cleaned_mixture_ot(n):
    cfg = problems.orszag_tang_two_fluid_2d(n)
    cfg["actions"]["divergence_cleaning"] = {"type": "elliptic", "projection": 1, "state": "field"}
    return problems.with_mixture(cfg, "ion", [lambda d: 0.5 + 0.3 * np.sin(d["x"]) * np.cos(d["y"])], [1.0, 3.0],
                                 [5 / 3, 1.4], [1.0, 2.0])


def test_mixture_charge_density_in_the_projection_oracle(oracle):
    """the elliptic action's charge density takes charge / mass of a cell from its mass fractions
    (HydroState::calc_current_and_charge, MFP_hydro.cpp:1720-1721): after the projection div D equals that density"""
    P = config.Problem(_cleaned_mixture_ot(32))
    plain = _cleaned_mixture_ot(32)
    del plain["actions"]["divergence_cleaning"]
    a, b = oracle.OracleSim(P), oracle.OracleSim(config.Problem(plain))
    dt = b.est_time_step()
    a.advance(dt)
    b.advance(dt)
    fs, si, se = (P.state_index[k] for k in ("field", "ion", "electron"))
    I, E = b.get_state(si), b.get_state(se)
    assert np.array_equal(a.get_state(si), I)                      # the projection leaves the fluids alone
    al = I[5] / I[0]
    n_over_rho = al / 1.0 + (1 - al) / 3.0                         # 1 / m of the mixture, written out for two species
    r_ion = ((al * 1.0 / 1.0 + (1 - al) * 2.0 / 3.0) / n_over_rho) * n_over_rho
    r = P.ref
    want = oracle.project_divergence_field(b.get_state(fs), [I[0], E[0]], [r_ion, -1.0 / 0.04], P.dx, P.dim,
                                           r["Larmor"] / (r["Debye"] ** 2 * r["lightspeed"]))
    got = a.get_state(fs)
    assert rel_linf(got, want) <= 1e-13
    const = oracle.project_divergence_field(b.get_state(fs), [I[0], E[0]], [1.0, -1.0 / 0.04], P.dx, P.dim,
                                            r["Larmor"] / (r["Debye"] ** 2 * r["lightspeed"]))
    assert rel_linf(got, const) > 1e-6                             # the charge / mass of species 0 alone is elsewhere


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [abi.FP_STRICT, abi.FP_FAST])
def test_mixture_with_elliptic_action_gpu(oracle, mode):
    """two-fluid run with mixture ions and the divergence projection after every step: as for the single-species
    case held to 1e-11 (the FFT round-off of the projection enters every step)"""
    P = config.Problem(_cleaned_mixture_ot(64), fp_mode=mode)
    ref = oracle.OracleSim(config.Problem(_cleaned_mixture_ot(64)))
    m = _gpu(P)
    dts_ref, dts = ref.run(5), m.run(5)
    np.testing.assert_allclose(dts, dts_ref, rtol=1e-12)
    for s in range(3):
        assert rel_linf(m.download(s), ref.get_state(s)) <= 1e-11, P.state_names[s]
    m.close()


@pytest.mark.gpu
def test_mixture_ions_in_collisions_beside_plasma5(oracle):
    """two-fluid Orszag-Tang, Strang: the ions a mixture of two kinds, the second of which collides with the electrons
    (collisions_rambo beside plasma5; stage-by-stage kernels).  Coulomb branch: pow(x, -1.5) of the device library
    against glibc's, hence 1e-13 in STRICT."""
    cfg = problems.orszag_tang_two_fluid_2d(24, collisions=True)
    problems.with_mixture(cfg, "ion", [lambda d: 0.4 + 0.2 * np.sin(d["x"])], [1.0, 2.0], [5 / 3, 1.4], [1.0, 2.0])
    cfg["actions"]["collisions"]["species"] = ["ion_1", "electron"]
    for mode, tol in ((abi.FP_STRICT, 1e-13), (abi.FP_FAST, 1e-12)):
        P = config.Problem(cfg, fp_mode=mode)
        ref = oracle.OracleSim(P)
        dts_ref = ref.run(4)
        m = _gpu(P)
        dts = m.run(4)
        assert m.last_advance_path() == 0
        np.testing.assert_allclose(dts, dts_ref, rtol=1e-12)
        for s in range(3):
            assert rel_linf(m.download(s), ref.get_state(s)) <= tol, (mode, P.state_names[s])
        # the collisions do something: the same run without them differs
        m.close()
    c2 = problems.orszag_tang_two_fluid_2d(24)
    problems.with_mixture(c2, "ion", [lambda d: 0.4 + 0.2 * np.sin(d["x"])], [1.0, 2.0], [5 / 3, 1.4], [1.0, 2.0])
    r2 = oracle.OracleSim(config.Problem(c2))
    r2.run(4)
    assert rel_linf(r2.get_state(0), ref.get_state(0)) > 1e-9


@pytest.mark.gpu
def test_mixture_refusals_of_the_library():
    """a collisions species that names no sub-component of its state: CMFP_ERR_INVALID, nothing silently approximated
    (a mixture in a collisions action itself is built: tests/test_gpu_actions.py::test_collisions_sub_component_parity)"""
    cfg = problems.orszag_tang_two_fluid_2d(16, collisions=True)
    problems.with_mixture(cfg, "ion", [0.5], [1.0, 2.0], [5 / 3, 1.4])
    with pytest.raises(ValueError):     # the default species are the state names: "ion" is no sub-component of ion_0, ion_1
        config.Problem(cfg).action_descs()         # ("Couldn't find sub-component", MFP_collisions.cpp:59)
    cfg["actions"]["collisions"]["species"] = ["ion_1", "electron"]
    P = config.Problem(cfg)
    descs = P.action_descs()
    for a in descs:
        if a.kind == abi.ACTION["collisions_rambo"]:
            a.species_component[0] = 2          # the ions have sub-components 0 and 1
    P.action_descs = lambda: descs
    with pytest.raises(lib.CmfpError) as e:
        _gpu(P)
    assert e.value.code == abi.ERR_INVALID
