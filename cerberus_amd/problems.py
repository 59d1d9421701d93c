"""The BASELINE.json configurations as problem dicts (see config.py).

Config 1 follows the reference's own input
Exec/testing/Braginskii_Riemann_2F/problem_definition.lua:1-235 with the
substitutions SURVEY.md §8(d) lists; configs 2-5 are authored here (the
reference ships no KH / Orszag-Tang / GEM / 3-D RMI input) from the same
section.  All initial data are analytic and deterministic.
"""
import math

import numpy as np


def brio_wu_two_fluid_1d(n=2048):
    # problem_definition.lua:8-58 (dimensional bookkeeping that fixes Larmor, Debye)
    mu_0_dim = 1.25663706e-6
    kb = 1.38064852e-23
    ref_lightspeed = 299792458.0
    q_ref = 1.60217662e-19
    m_ref = 1.6726219000e-27
    lightspeed = 2000
    ref_length = 1e-8
    ref_mass = m_ref
    ref_density = 1e31 * m_ref
    beta = 1
    mass_ratio = 100
    ref_velocity = ref_lightspeed / lightspeed
    n_ref = ref_density / ref_mass
    B_ref = math.sqrt(2 * mu_0_dim * n_ref * ref_mass * ref_velocity * ref_velocity / beta)
    ref_omega_c = q_ref * B_ref / ref_mass
    ref_larmor_dim = ref_velocity / ref_omega_c
    ref_skin_dim = ref_mass / (q_ref * math.sqrt(mu_0_dim * ref_density))
    ref_skin_nd = ref_skin_dim / ref_length
    Larmor = ref_larmor_dim / ref_length
    Debye = ref_skin_nd / lightspeed

    def step(A, B):
        return lambda dat: np.where(dat["x"] <= 0.0, A, B)     # :107-113

    hydro = lambda m, q, rs: {
        "type": "hydro",
        "gas": {"type": "thermally_perfect", "mass": m, "charge": q, "gamma": 5 / 3},
        "reconstruction": "vanLeer", "flux": "HLLE",
        "value": {"rho": step(1.0 * rs, 0.125 * rs), "x_vel": 0.0, "y_vel": 0.0, "z_vel": 0.0,
                  "p": step(0.5, 0.05)},
    }
    return {
        "dim": 1, "n_cell": [n], "prob_lo": [-2.0], "prob_hi": [2.0], "is_periodic": [0],
        "cfl": 0.25, "time_integration_scheme": "strang",
        "lightspeed": lightspeed, "Larmor": Larmor, "Debye": Debye,
        "states": {
            "ion": hydro(1.0, 1.0, 1.0),
            "electron": hydro(1.0 / mass_ratio, -1.0, 1.0 / mass_ratio),
            "field": {"type": "field", "reconstruction": "O6", "flux": "RankineHugoniot",
                      "value": {"x_B": 0.75, "y_B": step(1.0, -1.0), "z_B": 0.0}},
        },
        "actions": {
            "em_fluxes": {"type": "CTU", "corner_transport": True, "states": ["field"]},
            "hydro_fluxes": {"type": "CTU", "corner_transport": True, "states": ["ion", "electron"]},
            "plasma": {"type": "plasma5", "solver": "explicit", "states": ["ion", "electron", "field"]},
        },
    }


def kelvin_helmholtz_2d(n=2048, flux="HLLC", recon="minmod"):
    gam = 1.4
    band = lambda dat: np.abs(dat["y"] - 0.5) < 0.25
    return {
        "dim": 2, "n_cell": [n, n], "prob_lo": [0.0, 0.0], "prob_hi": [1.0, 1.0], "is_periodic": [1, 1],
        "cfl": 0.4, "time_integration_scheme": "strang", "lightspeed": 1.0,
        "states": {
            "fluid": {
                "type": "hydro",
                "gas": {"type": "thermally_perfect", "mass": 1.0, "charge": 0.0, "gamma": gam},
                "reconstruction": recon, "flux": flux,
                "value": {"rho": lambda d: np.where(band(d), 2.0, 1.0),
                          "x_vel": lambda d: np.where(band(d), 0.5, -0.5),
                          "y_vel": lambda d: 0.01 * np.sin(4 * math.pi * d["x"]),
                          "z_vel": 0.0, "p": 2.5},
            },
        },
        "actions": {"fluxes": {"type": "CTU", "corner_transport": True, "states": ["fluid"]}},
    }


def with_tracers(cfg, state, alphas):
    """Gives hydro state `state` of cfg len(alphas) passive scalars: a gas of len(alphas) + 1 species
    with equal mass, charge and gamma (n_tracers = n_species - 1, MFP_hydro_gas.H:16-17) and
    value.alpha = alphas (numbers or functions of the cell-centre dict)."""
    sd = cfg["states"][state]
    n = len(alphas) + 1
    gas = sd["gas"]
    for k in ("mass", "charge", "gamma"):
        gas[k] = [gas[k]] * n
    sd["value"]["alpha"] = list(alphas)
    return cfg


def with_mixture(cfg, state, alphas, masses, gammas, charges=None):
    """Makes hydro state `state` a mixture of len(alphas) + 1 species of the given masses, gammas and charges
    (ThermallyPerfectGas with mass_const / gamma_const / charge_const false, MFP_thermally_perfect_gas.cpp:32-34;
    charges default to the state's): value.alpha = mass fractions of all species but the last."""
    sd = cfg["states"][state]
    n = len(alphas) + 1
    if len(masses) != n or len(gammas) != n:
        raise ValueError("one mass and one gamma per species")
    gas = sd["gas"]
    q = gas["charge"]
    gas["mass"], gas["gamma"] = [float(m) for m in masses], [float(g) for g in gammas]
    gas["charge"] = [float(x) for x in charges] if charges is not None else [q[0] if isinstance(q, (list, tuple)) else q] * n
    if len(gas["charge"]) != n:
        raise ValueError("one charge per species")
    sd["value"]["alpha"] = list(alphas)
    return cfg


def shock_bubble_2d(nx=64, ny=32, dim=2, nz=8, recon="vanLeer", flux="HLLC", species=2):
    """A Mach-1.5 shock in air (gamma 1.4) running over a helium-like bubble (gamma 5/3, mass 0.138): the classic
    two-gas test of a mixture gas model.  Walls in y, outflow in x; `species` = 3 adds a heavy third gas in a band."""
    gam = 1.4
    M = 1.5
    p1, r1 = 1.0, 1.0
    p2 = p1 * (1 + 2 * gam / (gam + 1) * (M * M - 1))
    r2 = r1 * ((gam + 1) * M * M) / ((gam - 1) * M * M + 2)
    a1 = math.sqrt(gam * p1 / r1)
    u2 = M * a1 * (1 - r1 / r2)
    post = lambda d: d["x"] < 0.2
    R = lambda d: np.sqrt((d["x"] - 0.45) ** 2 + (d["y"] - 0.25) ** 2 + ((d["z"] - 0.0625) ** 2 if dim == 3 else 0.0))
    # smooth bubble edge: mass fraction of air
    air = lambda d: 0.5 * (1 + np.tanh((R(d) - 0.15) / 0.03))
    rho = lambda d: np.where(post(d), r2, r1 * (air(d) + (1 - air(d)) * 0.138))
    alphas = [air]
    masses, gammas = [1.0, 0.138], [gam, 5.0 / 3.0]
    if species == 3:
        band = lambda d: 0.3 * (np.abs(d["x"] - 0.75) < 0.08)
        alphas = [lambda d: air(d) - band(d), band]
        masses, gammas = [1.0, 2.0, 0.138], [gam, 1.1, 5.0 / 3.0]
    wall = {"lo": {"fill_hydro_bc": "slipwall"}, "hi": {"fill_hydro_bc": "slipwall"}}
    out = {"lo": {"fill_hydro_bc": "outflow"}, "hi": {"fill_hydro_bc": "outflow"}}
    cfg = {
        "dim": dim, "n_cell": [nx, ny, nz][:dim], "prob_lo": [0.0] * dim, "prob_hi": [1.0, 0.5, 0.125][:dim],
        "is_periodic": [0, 0, 1][:dim], "cfl": 0.4, "time_integration_scheme": "strang", "lightspeed": 1.0,
        "states": {
            "fluid": {
                "type": "hydro",
                "gas": {"type": "thermally_perfect", "mass": 1.0, "charge": 0.0, "gamma": gam},
                "reconstruction": recon, "flux": flux,
                "value": {"rho": rho, "x_vel": lambda d: np.where(post(d), u2, 0.0), "y_vel": 0.0, "z_vel": 0.0,
                          "p": lambda d: np.where(post(d), p2, p1)},
                "bc": {"x": out, "y": wall},
            },
        },
        "actions": {"fluxes": {"type": "CTU", "corner_transport": True, "states": ["fluid"]}},
    }
    if dim == 1:
        del cfg["states"]["fluid"]["bc"]["y"]
    return with_mixture(cfg, "fluid", alphas, masses, gammas)


def orszag_tang_two_fluid_2d(n=4096, solver="explicit", source="plasma5", collisions=False):
    gam = 5 / 3
    c = 20.0
    L = 2 * math.pi
    nd = gam * gam
    ux = lambda d: -np.sin(d["y"])
    uy = lambda d: np.sin(d["x"])
    Bx = lambda d: -np.sin(d["y"])
    By = lambda d: np.sin(2 * d["x"])
    # D = -(u x B)/c with u = (ux, uy, 0), B = (Bx, By, 0): only the z component is non-zero
    Dz = lambda d: -(ux(d) * By(d) - uy(d) * Bx(d)) / c
    hydro = lambda m, q: {
        "type": "hydro",
        "gas": {"type": "thermally_perfect", "mass": m, "charge": q, "gamma": gam},
        "reconstruction": "minmod", "flux": "HLLC",
        "value": {"rho": nd * m, "x_vel": ux, "y_vel": uy, "z_vel": 0.0, "p": gam / 2},
    }
    src = {"type": source, "states": ["ion", "electron", "field"]}
    if source == "plasma5":
        src["solver"] = solver
    cfg = {
        "dim": 2, "n_cell": [n, n], "prob_lo": [0.0, 0.0], "prob_hi": [L, L], "is_periodic": [1, 1],
        "cfl": 0.3, "time_integration_scheme": "strang",
        "lightspeed": c, "skin_depth": 0.2, "beta": 1.0,
        "states": {
            "ion": hydro(1.0, 1.0),
            "electron": hydro(1.0 / 25.0, -1.0),
            "field": {"type": "field", "reconstruction": "O6", "flux": "RankineHugoniot",
                      "div_transport": 1.0, "div_damping": 1.0,
                      "value": {"x_B": Bx, "y_B": By, "z_B": 0.0, "x_D": 0.0, "y_D": 0.0, "z_D": Dz}},
        },
        "actions": {
            "em_fluxes": {"type": "CTU", "corner_transport": True, "states": ["field"]},
            "hydro_fluxes": {"type": "CTU", "corner_transport": True, "states": ["ion", "electron"]},
            "plasma": src,
        },
    }
    if collisions:
        # ion-electron Coulomb collisions (CollisionsRambo) beside the Lorentz coupling; sorted before "plasma".
        # n0 = n_ref x_ref^3 sets the collisionality: nu ~ 1 / (n0 Debye^4)
        cfg["ref_density"] = 1.0e8
        cfg["actions"]["collisions"] = {"type": "collisions_rambo", "states": ["ion", "electron"]}
    return cfg


def collisions_relaxation(n=4, neutral=False):
    """Exec/testing/Collisions/Collisions.inputs: a carbon beam (A) relaxing on a carbon background (B)
    through CollisionsRambo alone (no flux action), RK2 with a forced time step.  neutral=True swaps B for
    a neutral gas so that the hard-sphere branch with the given cross section runs."""
    kB, ep0, q0 = 1.38064852e-23, 8.85418782e-12, 1.60217662e-19
    nd = 1e20 * 1e6
    ref_length, ref_mass, ref_temp = 1e-3, 1.67e-24 / 1e3, 11604525.00617
    ref_vel = math.sqrt(kB * ref_temp / ref_mass)
    beam_velocity = 6.55e7 * 1e-2 / ref_vel
    dD0 = math.sqrt((ep0 * kB * ref_temp) / (nd * q0 ** 2))
    gas = lambda q: {"type": "thermally_perfect", "mass": 12.0, "charge": q, "gamma": 5 / 3}
    return {
        "dim": 2, "n_cell": [n, n], "prob_lo": [-1.0, -1.0], "prob_hi": [1.0, 1.0], "is_periodic": [1, 1],
        "cfl": 0.5, "time_integration_scheme": "RK2", "force_dt": 0.0001,
        "ref_length": ref_length, "ref_mass": ref_mass, "ref_temp": ref_temp, "ref_density": nd * ref_mass,
        "Larmor": 1.0, "Debye": dD0 / ref_length,
        "states": {
            "A": {"type": "hydro", "gas": gas(6.0), "reconstruction": "centre", "flux": "HLLE",
                  "value": {"nd": 0.1, "x_vel": beam_velocity, "p": 0.05}},
            "B": {"type": "hydro", "gas": gas(0.0 if neutral else 6.0), "reconstruction": "centre", "flux": "HLLE",
                  "value": {"nd": 1.0, "p": 0.5}},
        },
        "actions": {
            "collisions": {"type": "collisions_rambo", "states": ["A", "B"],
                           "cross_sections": {"A": {"B": 1.0e-20 if neutral else 1.0, "C": 1.0}}},
        },
    }


def collisions_sub_component(n=4, neutral=False, passive=False):
    """collisions_relaxation with the beam state A made of two sub-components (gas.names = {A_C, A_He}) of which the
    action's `species` list names the second one: a carbon / helium mixture (different mass and charge), or with
    passive=True two labels of the same gas.  The mass fraction varies over the block."""
    cfg = collisions_relaxation(n, neutral)
    A = cfg["states"]["A"]
    if passive:
        A["gas"].update(mass=[12.0, 12.0], charge=[6.0, 6.0], gamma=[5 / 3, 5 / 3])
    else:
        A["gas"].update(mass=[12.0, 4.0], charge=[6.0, 2.0], gamma=[5 / 3, 1.4])
    A["gas"]["names"] = ["A_C", "A_He"]
    A["value"]["alpha"] = [lambda d: 0.4 + 0.3 * np.sin(np.pi * d["x"]) * np.cos(np.pi * d["y"])]
    act = cfg["actions"]["collisions"]
    act["species"] = ["A_He", "B"]
    act["cross_sections"] = {"A_He": {"B": 1.0e-20 if neutral else 1.0}}
    return cfg


def viscous_vortex_2d(n=384, recon="O6"):
    """Exec/testing/Viscous-Vortex (problem_definition.lua, vortex.inputs): a decaying Lamb-Oseen vortex
    (Munoz et al. 2013) in a viscous gas, PowerLaw viscosity with mu0 = 0.08, O6 + HLLC + CTU, `euler`
    integrator, periodic [-20, 20]^2.  The reference refines [-8, 8]^2 by 4 over 96^2 (dx = 0.104); n = 384
    is that resolution everywhere.  Its check.py accepts |v - v_theta| / v_theta <= 1 % for r <= 8 at t = 20."""
    from scipy.special import expi
    kB = 1.38064852e-23
    G, t0, mu0, rho0, p0 = 1.0, 10.0, 0.08, 1.0, 1.00548676
    nu = mu0 / rho0

    def P(r):   # pressure_definition.py
        return p0 - (G ** 2 * rho0 * ((-1 + np.exp(r ** 2 / (4. * t0 * nu))) ** 2 / (2. * np.exp(r ** 2 / (2. * t0 * nu)) * r ** 2)
                                      + (expi(-r ** 2 / (2. * t0 * nu)) - expi(-r ** 2 / (4. * t0 * nu))) / (4. * t0 * nu))) / (4. * np.pi ** 2)
    r_tab = np.linspace(1e-6, 30, 1000)      # the table the reference interpolates linearly (pressure.lua)
    p_tab = P(r_tab)
    radius = lambda d: np.sqrt(d["x"] ** 2 + d["y"] ** 2)
    vt = lambda r: G / (2 * math.pi * r) * (1 - np.exp(-r ** 2 / (4 * nu * t0)))
    return {
        "dim": 2, "n_cell": [n, n], "prob_lo": [-20.0, -20.0], "prob_hi": [20.0, 20.0], "is_periodic": [1, 1],
        "cfl": 0.5, "time_integration_scheme": "euler", "ref_temp": 1 / kB,
        "states": {
            "air": {"type": "hydro", "gas": {"type": "thermally_perfect", "mass": 1.0, "charge": 0.0, "gamma": 5 / 3},
                    "reconstruction": recon, "flux": "HLLC",
                    "viscosity": {"Pr": 0.71, "mu0": mu0, "T0": 1 / kB, "n": 1, "type": "PowerLaw"},
                    "value": {"rho": 1.0,
                              "x_vel": lambda d: vt(radius(d)) * d["y"] / radius(d),
                              "y_vel": lambda d: -vt(radius(d)) * d["x"] / radius(d),
                              "p": lambda d: np.interp(radius(d), r_tab, p_tab)}},
        },
        "actions": {"hydro_fluxes": {"type": "CTU", "corner_transport": True, "states": ["air"]}},
    }


def gem_reconnection_2d(nx=8192, ny=None):
    """GEM challenge (Birn et al. 2001) two-fluid set-up, SURVEY.md §8(d) config 4."""
    ny = nx // 2 if ny is None else ny
    Lx, Ly = 25.6, 12.8
    lam, B0, n0 = 0.5, 1.0, 1.0
    mi, me = 1.0, 1.0 / 25.0
    gam = 5 / 3
    c = 20.0
    Ti_Te = 5.0
    # pressure balance: n0 (Ti + Te) = B0^2 / 2  (beta = 1 units)
    Ttot = B0 * B0 / (2 * n0)
    Te = Ttot / (1 + Ti_Te)
    Ti = Ttot - Te
    psi0 = 0.1
    sech2 = lambda y: 1.0 / np.cosh(y / lam) ** 2
    ndens = lambda d: n0 * (0.2 + sech2(d["y"]))
    Bx = lambda d: (B0 * np.tanh(d["y"] / lam)
                    - psi0 * (math.pi / Ly) * np.cos(2 * math.pi * d["x"] / Lx) * np.sin(math.pi * d["y"] / Ly))
    By = lambda d: psi0 * (2 * math.pi / Lx) * np.sin(2 * math.pi * d["x"] / Lx) * np.cos(math.pi * d["y"] / Ly)
    wall = {"lo": {"fill_hydro_bc": "slipwall"}, "hi": {"fill_hydro_bc": "slipwall"}}
    fwall = {k: {"fill_D_bc": "asymmetry", "fill_B_bc": "asymmetry", "fill_phi_bc": "symmetry",
                 "fill_psi_bc": "symmetry", "fill_mu_bc": "symmetry", "fill_ep_bc": "symmetry"} for k in ("lo", "hi")}
    hydro = lambda m, q, T: {
        "type": "hydro",
        "gas": {"type": "thermally_perfect", "mass": m, "charge": q, "gamma": gam},
        "reconstruction": "minmod", "flux": "HLLC",
        "value": {"rho": lambda d: ndens(d) * m, "x_vel": 0.0, "y_vel": 0.0, "z_vel": 0.0,
                  "p": lambda d: ndens(d) * T},
        "bc": {"y": wall},
    }
    return {
        "dim": 2, "n_cell": [nx, ny], "prob_lo": [-Lx / 2, -Ly / 2], "prob_hi": [Lx / 2, Ly / 2],
        "is_periodic": [1, 0], "cfl": 0.3, "time_integration_scheme": "strang",
        "lightspeed": c, "skin_depth": 1.0, "beta": 1.0,
        "states": {
            "ion": hydro(mi, 1.0, Ti),
            "electron": hydro(me, -1.0, Te),
            "field": {"type": "field", "reconstruction": "O6", "flux": "RankineHugoniot",
                      "div_transport": 1.0, "div_damping": 1.0,
                      "value": {"x_B": Bx, "y_B": By, "z_B": 0.0}, "bc": {"y": fwall}},
        },
        "actions": {
            "em_fluxes": {"type": "CTU", "corner_transport": True, "states": ["field"]},
            "hydro_fluxes": {"type": "CTU", "corner_transport": True, "states": ["ion", "electron"]},
            "plasma": {"type": "plasma5", "solver": "explicit", "states": ["ion", "electron", "field"]},
        },
    }


def richtmyer_meshkov_3d(n=512, ny=None, nz=None):
    """3-D extension of Exec/testing/Thermal-RMI:
    Mach-2 shock at x = -0.2, interface x = 0.1 cos(2 pi y) cos(2 pi z), density
    1 -> 3, m_e/m_i = 0.01, without tracers / particles / elliptic."""
    ny = n if ny is None else ny
    nz = n if nz is None else nz
    gam = 5 / 3
    c, skin, beta = 20.0, 1.0, 1.0
    mach = 2.0
    p1, rho1 = 0.5, 1.0
    a1 = math.sqrt(gam * p1 / rho1)
    # Rankine-Hugoniot post-shock state for a shock moving in +x into (rho1, p1, u = 0)
    p2 = p1 * (1 + 2 * gam / (gam + 1) * (mach * mach - 1))
    rho2 = rho1 * ((gam + 1) * mach * mach) / ((gam - 1) * mach * mach + 2)
    u2 = mach * a1 * (1 - rho1 / rho2)
    width = 0.01
    shock_x = -0.2

    def interface(d):
        return 0.1 * np.cos(2 * math.pi * d["y"]) * np.cos(2 * math.pi * d["z"])

    def rho_nd(d):
        x = d["x"]
        light_heavy = 1.0 + 0.5 * (3.0 - 1.0) * (1 + np.tanh((x - interface(d)) / width))
        return np.where(x < shock_x, rho2, light_heavy)

    pres = lambda d: np.where(d["x"] < shock_x, p2, p1)
    uvel = lambda d: np.where(d["x"] < shock_x, u2, 0.0)
    hydro = lambda m, q: {
        "type": "hydro",
        "gas": {"type": "thermally_perfect", "mass": m, "charge": q, "gamma": gam},
        "reconstruction": "minmod", "flux": "HLLC",
        "value": {"rho": lambda d: rho_nd(d) * m, "x_vel": uvel, "y_vel": 0.0, "z_vel": 0.0,
                  "p": lambda d: 0.5 * pres(d)},
    }
    return {
        "dim": 3, "n_cell": [n, ny, nz], "prob_lo": [-0.5, 0.0, 0.0],
        "prob_hi": [-0.5 + 1.0 * n / max(ny, 1), 1.0, 1.0 * nz / max(ny, 1)],
        "is_periodic": [0, 1, 1], "cfl": 0.3, "time_integration_scheme": "strang",
        "lightspeed": c, "skin_depth": skin, "beta": beta,
        "states": {
            "ion": hydro(1.0, 1.0),
            "electron": hydro(0.01, -1.0),
            "field": {"type": "field", "reconstruction": "O6", "flux": "RankineHugoniot",
                      "div_transport": 1.0, "div_damping": 1.0, "value": {}},
        },
        "actions": {
            "em_fluxes": {"type": "CTU", "corner_transport": True, "states": ["field"]},
            "hydro_fluxes": {"type": "CTU", "corner_transport": True, "states": ["ion", "electron"]},
            "plasma": {"type": "plasma5", "solver": "explicit", "states": ["ion", "electron", "field"]},
        },
    }


def cavity_mode_2d(n=32, m=1, l=1, Lx=1.0, Ly=1.0, recon="O6"):
    """A TM_ml eigenmode of a rectangular cavity with perfectly conducting walls, with the option set of the
    reference's field-only example Exec/testing/Waveguide/Waveguide.inputs (lightspeed = 1, cfl = 0.5, `euler`,
    field O6 + RankineHugoniot, CTU with corner transport).  That example builds its conductor out of embedded
    boundaries (out of scope); on the uniform grid the conducting wall is the `asymmetry` field boundary
    (MFP_bc.cpp:99-109: tangential D and normal B odd, normal D and tangential B even).  Analytic solution:
    D_z = sin(kx x) sin(ky y) cos(w t), B_x = -(c ky / w) sin(kx x) cos(ky y) sin(w t),
    B_y = (c kx / w) cos(kx x) sin(ky y) sin(w t), w = c sqrt(kx^2 + ky^2), kx = m pi / Lx, ky = l pi / Ly."""
    kx, ky = m * math.pi / Lx, l * math.pi / Ly
    wall = {k: {"fill_D_bc": "asymmetry", "fill_B_bc": "asymmetry", "fill_phi_bc": "symmetry", "fill_psi_bc": "symmetry",
                "fill_mu_bc": "symmetry", "fill_ep_bc": "symmetry"} for k in ("lo", "hi")}
    return {
        "dim": 2, "n_cell": [n, max(2, int(round(n * Ly / Lx)))], "prob_lo": [0.0, 0.0], "prob_hi": [Lx, Ly],
        "is_periodic": [0, 0], "cfl": 0.5, "time_integration_scheme": "euler", "lightspeed": 1.0,
        "states": {"field": {"type": "field", "reconstruction": recon, "flux": "RankineHugoniot",
                             "value": {"z_D": lambda d: np.sin(kx * d["x"]) * np.sin(ky * d["y"])},
                             "bc": {"x": wall, "y": wall}}},
        "actions": {"fluxes": {"type": "CTU", "corner_transport": True, "states": ["field"]}},
        "_mode": {"kx": kx, "ky": ky, "omega": math.sqrt(kx * kx + ky * ky)},
    }


def uniform_plasma_2d(n=8, source="plasma5", u0=1e-3, B0=0.0, mass_ratio=25.0, lightspeed=20.0, skin_depth=0.2, beta=1.0):
    """A uniform, periodic two-fluid plasma: the flux update is exactly zero, so only the source action acts.
    plasma5 with a small electron drift u0 along x (no magnetic field): Langmuir oscillation of the current at
    w_p^2 = sum_s rho_s (q_s / m_s)^2 / Debye^2.  Lorentz with a uniform B_z = B0: each species gyrates at
    w_c = |q_s / m_s| B0 / Larmor.  Both frequencies are the ones the reference's own dt limit is built from
    (MFP_plasma5.cpp:549-555) and follow from its source terms (:123-129, 213-235)."""
    gam = 5 / 3
    hydro = lambda m, q, u: {
        "type": "hydro", "gas": {"type": "thermally_perfect", "mass": m, "charge": q, "gamma": gam},
        "reconstruction": "minmod", "flux": "HLLC",
        "value": {"rho": 1.0 * m, "x_vel": u, "y_vel": 0.0, "z_vel": 0.0, "p": 0.5}}
    src = {"type": source, "states": ["ion", "electron", "field"]}
    if source == "plasma5":
        src["solver"] = "explicit"
    return {
        "dim": 2, "n_cell": [n, n], "prob_lo": [0.0, 0.0], "prob_hi": [1.0, 1.0], "is_periodic": [1, 1],
        "cfl": 0.3, "time_integration_scheme": "strang", "lightspeed": lightspeed, "skin_depth": skin_depth, "beta": beta,
        "states": {
            "ion": hydro(1.0, 1.0, u0 if source == "Lorentz" else 0.0),
            "electron": hydro(1.0 / mass_ratio, -1.0, u0),
            "field": {"type": "field", "reconstruction": "O6", "flux": "RankineHugoniot", "value": {"z_B": B0}},
        },
        "actions": {
            "em_fluxes": {"type": "CTU", "corner_transport": True, "states": ["field"]},
            "hydro_fluxes": {"type": "CTU", "corner_transport": True, "states": ["ion", "electron"]},
            "plasma": src,
        },
    }


WORKLOADS = {
    "brio_wu_1d": brio_wu_two_fluid_1d,
    "kh_2d": kelvin_helmholtz_2d,
    "ot2f_2d": orszag_tang_two_fluid_2d,
    "gem_2d": gem_reconnection_2d,
    "rmi_3d": richtmyer_meshkov_3d,
    "collisions": collisions_relaxation,
    "viscous_vortex": viscous_vortex_2d,
}
