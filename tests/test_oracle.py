"""The CPU oracle against (a) the known-answer vectors the survey recorded from
the reference's own leaf sources (tests/golden/survey_kat.json),
(b) analytic solutions in the style of the reference's Exec/testing/*/check.py,
(c) structural properties the reference has by construction (independence of the
box decomposition, conservation, periodic symmetry)."""
import json
import math
import os

import numpy as np
import pytest

from cerberus_amd import abi, config, problems
from helpers import rel_linf, small_cases, survey_cases

HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "survey_kat.json")))
col = lambda v: np.array(v, dtype=float)[:, None]


# ---- (a) known-answer vectors: bit-exact -----------------------------------
def test_kat_hllc_hlle(oracle):
    assert np.array_equal(oracle.riemann(0, col(KAT["hydro_L"]), col(KAT["hydro_R"]))[:, 0], KAT["HLLC_F"])
    assert np.array_equal(oracle.riemann(1, col(KAT["hydro_L"]), col(KAT["hydro_R"]))[:, 0], KAT["HLLE_F"])


@pytest.mark.parametrize("name,r,key", [("minmod", 1, "stencil3"), ("vanLeer", 2, "stencil3"), ("MC", 3, "stencil3"),
                                         ("O6", 5, "stencil7"), ("WENO", 6, "stencil7")])
def test_kat_reconstruction(oracle, name, r, key):
    s, lo, hi = oracle.reconstruct(r, col(KAT[key]))
    assert [lo[0], hi[0]] == KAT[name]


def test_kat_field_rh(oracle):
    F = oracle.riemann(10, col(KAT["field_L"]), col(KAT["field_R"]), c0=KAT["field_c0"], ch=KAT["field_ch"])[:, 0]
    assert np.array_equal(F, KAT["field_F"])


def test_kat_gas(oracle):
    g = KAT["gas"]
    Q, s = oracle.cons2prim(g["mass"], g["gamma"], col(KAT["gas_U"]))
    assert np.array_equal(Q[:, 0], KAT["gas_Q"])
    assert list(s[:2, 0]) == KAT["gas_speeds_xy"]
    U = oracle.prim2cons(g["mass"], g["gamma"], Q)
    np.testing.assert_allclose(U[:, 0], KAT["gas_U"], rtol=2e-16)


def implicit_matrix(im):
    Ns, N = im["Ns"], 3 * (im["Ns"] + 1)
    dt, L, D, c, B = im["dt"], im["Larmor"], im["Debye"], im["c"], im["B"]
    A, b = np.eye(N), np.zeros(N)
    for n, (r, rho, m) in enumerate(zip(im["r"], im["rho"], [im["m_ion"], im["m_ele"]])):
        tB, tC, mc = -dt * r * rho * c / L, dt * r * L / (c * D * D), dt * r / L   # MFP_plasma5.cpp:373-375
        A[3 * n + 0, 3 * n + 1], A[3 * n + 0, 3 * n + 2] = -B[2] * mc, B[1] * mc
        A[3 * n + 1, 3 * n + 0], A[3 * n + 1, 3 * n + 2] = B[2] * mc, -B[0] * mc
        A[3 * n + 2, 3 * n + 0], A[3 * n + 2, 3 * n + 1] = -B[1] * mc, B[0] * mc
        for i in range(3):
            A[3 * n + i, 3 * Ns + i], A[3 * Ns + i, 3 * n + i] = tB, tC
        b[3 * n:3 * n + 3] = m
    b[3 * Ns:] = im["E"]
    return A, b


def test_kat_implicit_system(oracle):
    A, b = implicit_matrix(KAT["implicit"])
    x = oracle.colpiv_solve(A, b)
    sol = np.array(KAT["implicit"]["solution"])
    assert np.abs(x - sol).max() / np.abs(sol).max() < 1e-13


# ---- (b) analytic: Sod shock tube (exact Riemann solution, Toro ch. 4) -------
def sod_exact(x, t, gam=1.4, left=(1.0, 0.0, 1.0), right=(0.125, 0.0, 0.1)):
    rl, ul, pl = left
    rr, ur, pr = right
    al, ar = math.sqrt(gam * pl / rl), math.sqrt(gam * pr / rr)

    def f(p, r, pk, a):
        if p > pk:
            A, B = 2 / ((gam + 1) * r), (gam - 1) / (gam + 1) * pk
            return (p - pk) * math.sqrt(A / (p + B))
        return 2 * a / (gam - 1) * ((p / pk) ** ((gam - 1) / (2 * gam)) - 1)

    lo, hi = 1e-8, 10.0
    for _ in range(200):
        mid = 0.5 * (lo + hi)
        if f(mid, rl, pl, al) + f(mid, rr, pr, ar) + ur - ul > 0:
            hi = mid
        else:
            lo = mid
    ps = 0.5 * (lo + hi)
    us = 0.5 * (ul + ur) + 0.5 * (f(ps, rr, pr, ar) - f(ps, rl, pl, al))
    rsl = rl * (ps / pl) ** (1 / gam)                    # left rarefaction
    asl = al * (ps / pl) ** ((gam - 1) / (2 * gam))
    rsr = rr * ((ps / pr + (gam - 1) / (gam + 1)) / ((gam - 1) / (gam + 1) * ps / pr + 1))   # right shock
    S = ur + ar * math.sqrt((gam + 1) / (2 * gam) * ps / pr + (gam - 1) / (2 * gam))
    rho = np.empty_like(x)
    for i, xi in enumerate(x):
        s = xi / t
        if s < ul - al:
            rho[i] = rl
        elif s < us - asl:
            rho[i] = rl * (2 / (gam + 1) + (gam - 1) / ((gam + 1) * al) * (ul - s)) ** (2 / (gam - 1))
        elif s < us:
            rho[i] = rsl
        elif s < S:
            rho[i] = rsr
        else:
            rho[i] = rr
    return rho


def sod_cfg(n, flux="HLLC", recon="minmod"):
    step = lambda a, b: (lambda d: np.where(d["x"] <= 0.0, a, b))
    return {"dim": 1, "n_cell": [n], "prob_lo": [-0.5], "prob_hi": [0.5], "is_periodic": [0], "cfl": 0.5,
            "time_integration_scheme": "strang", "lightspeed": 1.0,
            "states": {"gas": {"type": "hydro", "gas": {"type": "thermally_perfect", "mass": 1.0, "charge": 0.0,
                                                         "gamma": 1.4},
                               "reconstruction": recon, "flux": flux,
                               "value": {"rho": step(1.0, 0.125), "x_vel": 0.0, "y_vel": 0.0, "z_vel": 0.0,
                                         "p": step(1.0, 0.1)}}},
            "actions": {"flux": {"type": "CTU", "corner_transport": True, "states": ["gas"]}}}


@pytest.mark.parametrize("flux", ["HLLC", "HLLE"])
def test_sod_against_exact_solution(oracle, flux):
    n = 400
    P = config.Problem(sod_cfg(n, flux))
    sim = oracle.OracleSim(P)
    t, dt = 0.0, sim.est_time_step()
    while t < 0.2 - 1e-12:
        dt = min(dt, 0.2 - t)
        sim.advance(dt)
        t += dt
        dt = min(sim.est_time_step(), 1.1 * dt)
    rho = sim.get_state(0)[0, 0, 0]
    exact = sod_exact(P.cell_centres()["x"][0, 0], 0.2)
    assert np.abs(rho - exact).mean() < 4e-3          # L1 error, second-order MUSCL at n = 400


# ---- (c) structural properties -----------------------------------------------
@pytest.mark.parametrize("case", range(5))
def test_independent_of_box_decomposition(oracle, case):
    """The reference runs the same stencils on every box of its BoxArray
    (amr.max_grid_size); results cannot depend on how the level is chopped."""
    name, cfg, steps = small_cases()[case]
    P = config.Problem(cfg)
    a, b = oracle.OracleSim(P), oracle.OracleSim(P, max_grid_size=8)
    assert a.run(steps) == b.run(steps)
    for s in range(len(P.state_names)):
        assert np.array_equal(a.get_state(s), b.get_state(s)), (name, s)


def test_periodic_conservation(oracle):
    P = config.Problem(problems.kelvin_helmholtz_2d(32))
    sim = oracle.OracleSim(P)
    U0 = sim.get_state(0)
    sim.run(10)
    U1 = sim.get_state(0)
    for c in range(5):
        assert abs(U1[c].sum() - U0[c].sum()) <= 1e-12 * max(1.0, abs(U0[c]).sum())


def test_ghost_fill_rules(oracle):
    """fab_filcc semantics used for FillPatch: foextrap copies the first interior
    cell, reflect_even mirrors, reflect_odd mirrors with a sign flip, periodic wraps;
    corners inherit through the dimension-by-dimension sweeps."""
    P = config.Problem(problems.gem_reconnection_2d(16, 8))   # periodic x, walls in y
    sim = oracle.OracleSim(P)
    s = P.state_index["ion"]
    U = sim.get_state(s)[:, 0]
    G = sim.fill_patch(s)[:, 0]
    ng = 2
    assert np.array_equal(G[:, ng:-ng, ng:-ng], U)
    assert np.array_equal(G[:, ng:-ng, :ng], U[:, :, -ng:])                 # periodic x
    assert np.array_equal(G[0, ng - 1, ng:-ng], U[0, 0])                    # rho: reflect_even
    assert np.array_equal(G[0, 0, ng:-ng], U[0, 1])
    assert np.array_equal(G[2, ng - 1, ng:-ng], -U[2, 0])                   # y momentum: reflect_odd
    assert np.array_equal(G[1, -1, ng:-ng], U[1, -2])                       # x momentum (tangential, slipwall): even
    assert np.array_equal(G[2, ng - 1, :ng], -U[2, 0, -ng:])                # corner: wrap in x of the reflected row


def test_implicit_and_explicit_sources_agree_for_small_dt(oracle):
    Pe = config.Problem(problems.orszag_tang_two_fluid_2d(16, solver="explicit"))
    Pi = config.Problem(problems.orszag_tang_two_fluid_2d(16, solver="implicit"))
    a, b = oracle.OracleSim(Pe), oracle.OracleSim(Pi)
    # the Orszag-Tang start is force-free (E = -u x B / c): kick the electric field out of balance
    f = Pe.state_index["field"]
    F = a.get_state(f)
    x = Pe.cell_centres()
    F[0] += 0.01 * np.sin(x["x"])
    F[1] += 0.02 * np.cos(x["y"])
    F[2] += 0.015
    a.set_state(f, F)
    b.set_state(f, F)
    dt = 1e-6
    s = Pe.state_index["ion"]
    # plasma action is index 2; first-order agreement of the momentum source
    da, db = a.action_dU(2, dt, s), b.action_dU(2, dt, s)
    assert np.abs(da[1:4] - db[1:4]).max() < 1e-3 * np.abs(da[1:4]).max()


# ---- collisions_rambo (no reference fixture: parity unpinned; physics properties) --------
@pytest.mark.parametrize("neutral", [False, True])
def test_collisions_on_a_sub_component(oracle, neutral):
    """A species of the action that is one sub-component of a mixture state (gas.names + the action's `species`
    list, MFP_collisions.cpp:43-60,160-257): what the beam loses the background gains, the sub-components' densities
    do not move (S = 0), and two labels of one gas (passive tracer) give the five hydro components of the same
    problem without labels when the labelled share is the whole state."""
    P = config.Problem(problems.collisions_sub_component(4, neutral))
    assert list(P.action_descs()[0].species_component)[:2] == [1, 0]
    sim = oracle.OracleSim(P)
    U0 = [sim.get_state(s).copy() for s in range(2)]
    for _ in range(40):
        sim.advance(1e-4)
    U1 = [sim.get_state(s) for s in range(2)]
    for c in (1, 2, 3, 4):
        t0, t1 = sum(u[c].sum() for u in U0), sum(u[c].sum() for u in U1)
        assert abs(t1 - t0) <= 1e-14 * max(abs(t0), 1.0), c
    assert np.array_equal(U1[0][0], U0[0][0]) and np.allclose(U1[0][5], U0[0][5], rtol=1e-15, atol=0)
    assert U1[0][1].mean() < 0.95 * U0[0][1].mean()          # the beam slows down on the background
    # the whole state labelled as the participating sub-component = the problem without labels, to rounding
    cfg = problems.collisions_sub_component(4, neutral, passive=True)
    cfg["states"]["A"]["value"]["alpha"] = [0.0]              # all of A is its second sub-component (A_He)
    a = oracle.OracleSim(config.Problem(cfg))
    b = oracle.OracleSim(config.Problem(problems.collisions_relaxation(4, neutral)))
    for _ in range(10):
        a.advance(1e-4)
        b.advance(1e-4)
    for s in range(2):
        assert np.allclose(a.get_state(s)[:5], b.get_state(s), rtol=1e-13, atol=0)
    # a species that is not a sub-component of the state at its own position: the reference reads the wrong
    # state's data there (MFP_collisions.cpp:181-190); refused by the configuration layer
    cfg = problems.collisions_sub_component(4, neutral)
    cfg["actions"]["collisions"]["species"] = ["B", "A_He"]
    with pytest.raises(NotImplementedError):
        config.Problem(cfg).action_descs()
    cfg["actions"]["collisions"]["species"] = ["A_He", "nobody"]
    with pytest.raises(ValueError):
        config.Problem(cfg).action_descs()


@pytest.mark.parametrize("neutral", [False, True])
def test_collisions_conserve_and_equilibrate(oracle, neutral):
    """Exec/testing/Collisions: the beam and the background exchange momentum and energy only
    with each other (R and Q enter with opposite signs, MFP_collisions_rambo.cpp:112-123) and
    relax to a common velocity and temperature."""
    from cerberus_amd import config, problems
    P = config.Problem(problems.collisions_relaxation(4, neutral))
    sim = oracle.OracleSim(P)
    U0 = [sim.get_state(s).copy() for s in range(2)]
    for _ in range(300):
        sim.advance(1e-4)
    U1 = [sim.get_state(s) for s in range(2)]
    tot0, tot1 = U0[0] + U0[1], U1[0] + U1[1]
    assert np.abs(tot1 - tot0).max() <= 1e-12 * np.abs(tot0).max()
    assert np.array_equal(U1[0][0], U0[0][0]) and np.array_equal(U1[1][0], U0[1][0])   # S = 0: densities untouched

    def uT(U):
        rho, mx, E = U[0], U[1], U[4]
        return mx / rho, (E - 0.5 * mx * mx / rho) * (2 / 3) / rho * 12.0
    (ua, Ta), (ub, Tb) = uT(U1[0]), uT(U1[1])
    assert np.abs(ua - ub).max() < 1e-2 * np.abs(uT(U0[0])[0]).max()
    assert np.abs(Ta - Tb).max() < 2e-2 * Ta.max()
    assert Ta.min() > uT(U0[0])[1].max()          # the beam's drift energy became heat


def test_force_dt_overrides_the_cfl_estimate(oracle):
    """MFP::estTimeStep returns force_dt before anything else (MFP_timestep.cpp:8): the shipped
    collisions problem (force_dt = 1e-4) steps with exactly that value, growth cap included."""
    from cerberus_amd import config, problems
    P = config.Problem(problems.collisions_relaxation(4, False))
    assert P.force_dt == 1e-4
    sim = oracle.OracleSim(P)
    assert sim.est_time_step() == 1e-4
    assert sim.run(5) == [1e-4] * 5
    cfg = problems.collisions_relaxation(4, False)
    cfg["force_dt"] = 0.0
    assert oracle.OracleSim(config.Problem(cfg)).est_time_step() != 1e-4


# ---- passive scalars (tracers) ------------------------------------------------------------
def _kh_tracers(flux="HLLC", n=32):
    cfg = problems.kelvin_helmholtz_2d(n, flux=flux, recon="vanLeer")
    band = lambda d: (np.abs(d["y"] - 0.5) < 0.25).astype(float)
    return problems.with_tracers(cfg, "fluid", [band, 1.0])


@pytest.mark.parametrize("flux", ["HLLC", "HLLE"])
def test_tracers_are_passive_conserved_and_keep_a_constant(oracle, flux):
    """hydro state with two tracers (three species of equal mass / charge / gamma): the five hydro
    components evolve exactly as without tracers, rho * alpha is conserved on the periodic domain,
    alpha = 1 stays 1 (its flux is the mass flux), 0 <= alpha <= 1 for the band marker."""
    P = config.Problem(_kh_tracers(flux))
    assert P.n_comp(0) == 7 and P.state_descs()[0].n_tracers == 2
    P0 = config.Problem(problems.kelvin_helmholtz_2d(32, flux=flux, recon="vanLeer"))
    a, b = oracle.OracleSim(P), oracle.OracleSim(P0)
    U0 = a.get_state(0).copy()
    assert a.run(12) == b.run(12)
    U, V = a.get_state(0), b.get_state(0)
    assert np.array_equal(U[:5], V)
    for c in (5, 6):
        assert abs(U[c].sum() - U0[c].sum()) <= 1e-12 * np.abs(U0[c]).sum()
    np.testing.assert_allclose(U[6] / U[0], 1.0, rtol=0, atol=1e-13)
    al = U[5] / U[0]
    assert al.min() > -1e-12 and al.max() < 1 + 1e-12
    assert np.abs(al - U0[5] / U0[0]).max() > 1e-3        # and it did move
    c, d = oracle.OracleSim(P), oracle.OracleSim(P, max_grid_size=8)
    c.run(4)
    d.run(4)
    assert np.array_equal(c.get_state(0), d.get_state(0))


# ---- Dirichlet (inflow) faces -----------------------------------------------------------------
def inflow_cfg(n=48, ny=16, scheme="strang"):
    """uniform supersonic stream entering at x-lo with a y-dependent density profile imposed through
    bc.x.lo (fill_hydro_bc = 'inflow': every primitive given), leaving through an outflow at x-hi;
    a moving slip wall in y with a prescribed tangential velocity value (the HartmannFlow input's
    `y_vel = v_plate` on a wall face, here x_vel)."""
    rho_in = lambda d: 1.0 + 0.2 * np.sin(2 * math.pi * d["y"])
    inflow = {"fill_hydro_bc": "inflow", "rho": rho_in, "x_vel": 3.0, "y_vel": 0.0, "z_vel": 0.0, "p": 1.0,
              "T": 1.0, "gamma": 1.4, "cp": 3.5}
    cfg = problems.kelvin_helmholtz_2d(n, flux="HLLC", recon="vanLeer")
    cfg.update({"n_cell": [n, ny], "prob_hi": [1.0, ny / n], "is_periodic": [0, 0], "time_integration_scheme": scheme})
    st = cfg["states"]["fluid"]
    st["value"] = {"rho": 1.0, "x_vel": 3.0, "y_vel": 0.0, "z_vel": 0.0, "p": 1.0}
    st["bc"] = {"x": {"lo": inflow, "hi": {"fill_hydro_bc": "outflow"}},
                "y": {"lo": {"fill_hydro_bc": "slipwall"}, "hi": {"fill_hydro_bc": "slipwall", "x_vel": 3.0}}}
    return cfg


def test_inflow_face_imposes_the_profile(oracle):
    """supersonic inflow: after the stream has crossed a third of the domain the imposed density
    profile has been carried in (first cells carry the face values), the far field is untouched"""
    P = config.Problem(inflow_cfg())
    sim = oracle.OracleSim(P)
    t = 0.0
    while t < 0.1:
        dt = sim.est_time_step()
        sim.advance(dt)
        t += dt
    U = sim.get_state(0)
    y = P.prob_lo[1] + (np.arange(16) + 0.5) * P.dx[1]
    prof = 1.0 + 0.2 * np.sin(2 * math.pi * y)
    assert np.abs(U[0, 0, 2:-2, 0] - prof[2:-2]).max() < 2e-2          # the inflow column carries the profile
    assert np.abs(U[0, 0, :, -4:] - 1.0).max() < 1e-12                 # nothing has reached the outflow yet
    assert np.abs(U[0, 0, :, 10] - 1.0).max() > 0.05                   # the profile has moved in (x = 0.22 < 0.3)
    with pytest.raises(ValueError):                                     # inflow needs every primitive (MFP_hydro.cpp:361-365)
        cfg = inflow_cfg()
        del cfg["states"]["fluid"]["bc"]["x"]["lo"]["p"]
        config.Problem(cfg).face_entries(0)


# ---- committed N-step vectors (tests/golden/oracle_nstep.json, tools/make_golden_nstep.py) ----------
ALL_GOLDEN = small_cases() + survey_cases()


@pytest.mark.parametrize("name,cfg,steps", ALL_GOLDEN, ids=[c[0] for c in ALL_GOLDEN])
def test_oracle_reproduces_committed_nstep_vectors(oracle, name, cfg, steps):
    """the restatement's N-step result (dt sequence and every conserved value, bit for bit) is the one
    committed with the repository: orchestration, ghost fill and integrator order cannot drift silently"""
    import hashlib
    G = json.load(open(os.path.join(HERE, "golden", "oracle_nstep.json")))[name]
    P = config.Problem(cfg, fp_mode=abi.FP_STRICT)
    # the survey-size cases run chopped into 32-cell boxes (as the reference's inputs do; one thread per box)
    sim = oracle.OracleSim(P, max_grid_size=32 if "@" in name else 0)
    dts = sim.run(G["steps"])
    assert [float(v).hex() for v in dts] == G["dts"]
    for s, sname in enumerate(P.state_names):
        a = np.ascontiguousarray(sim.get_state(s), dtype=np.float64)
        assert list(a.shape) == G["states"][sname]["shape"]
        assert hashlib.sha256(a.tobytes()).hexdigest() == G["states"][sname]["sha256"], (name, sname)
