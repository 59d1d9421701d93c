"""Problem definition -> C-ABI descriptors.

The reference reads an AMReX `inputs` file plus a Lua script
(Source/MFP_config.cpp:40-356, defaults Source/default.lua:12-52).  Lua is not
available in this image, so the same tables are given as a Python dict with the
same keys: `states.<name>.{type,gas,reconstruction,flux,
value,bc,div_transport,div_damping}`, `actions.<name>.{type,corner_transport,
solver,states}`, the globals `cfl, time_integration_scheme, lightspeed, beta,
skin_depth, Larmor, Debye` and the geometry keys of the inputs file.  Value
entries are numbers or vectorised callables f(dat) with dat['x'|'y'|'z'].
"""
import math

import numpy as np

from . import abi, bc

DEFAULTS = dict(lightspeed=-1.0, beta=1.0, skin_depth=1.0, Larmor=-1.0, Debye=-1.0, cfl=0.0,
                time_integration_scheme="strang", force_dt=0.0,
                ref_length=1.0, ref_density=1.0, ref_mass=1.0, ref_temp=-1.0, ref_lightspeed=299792458.0)  # default.lua:12-16

HYDRO_PRIM = ["rho", "x_vel", "y_vel", "z_vel", "p", "T", "gamma", "cp"]   # MFP_hydro.cpp:21-30
FIELD_CONS = ["x_D", "y_D", "z_D", "x_B", "y_B", "z_B", "phi", "psi", "mu", "ep"]  # MFP_field.cpp:20-31
FIELD_DEFAULT = {"phi": 0.0, "psi": 0.0, "mu": 1.0, "ep": 1.0}            # MFP_field.cpp:180-185


def update_ref(cfg):
    """MFP::update_ref (MFP_config.cpp:374-457): lightspeed, Larmor, Debye."""
    g = dict(DEFAULTS)
    g.update({k: cfg[k] for k in DEFAULTS if k in cfg})
    x_ref, rho_ref, m_ref, T_ref, c_ref = g["ref_length"], g["ref_density"], g["ref_mass"], g["ref_temp"], g["ref_lightspeed"]
    n0 = (rho_ref / m_ref) * x_ref * x_ref * x_ref   # :390-392
    kB = 1.38064852e-23
    c = g["lightspeed"]
    if T_ref > 0 and c > 0:
        raise ValueError("ref_temp and lightspeed both set, only set one")   # :398
    if T_ref > 0:
        u_ref = math.sqrt(kB * T_ref / m_ref)         # :400-404
        c = c_ref / u_ref
    elif c > 0:
        u_ref = c_ref / c                              # :405-409
        T_ref = ((c_ref * c_ref) / (c * c)) * (m_ref / kB)
    else:
        raise ValueError("Either ref_temp or lightspeed need to be set, only set one")
    Larmor, Debye, beta, skin = g["Larmor"], g["Debye"], g["beta"], g["skin_depth"]
    if Larmor > 0 and Debye > 0:                    # :420-421, case 1
        skin = Debye * c
        beta = 2 * (Larmor / skin) * (Larmor / skin)
    elif beta > 0 and skin > 0:                     # case 2 :435-438
        Debye = skin / c
        Larmor = math.sqrt(beta / 2.0) * skin
    else:
        raise ValueError("plasma parameters not set")
    return dict(lightspeed=c, Larmor=Larmor, Debye=Debye, beta=beta, skin_depth=skin, n0=n0, x_ref=x_ref,
                rho_ref=rho_ref, u_ref=u_ref, T_ref=T_ref)


def gas_n_tracers(gas):
    """HydroGas::n_tracers = n_species - 1 (MFP_hydro_gas.H:16-17): the length of the mass list."""
    m = gas["mass"]
    return len(m) - 1 if isinstance(m, (list, tuple)) else 0


def gas_list(gas, key):
    """mass / charge / gamma of every species of the gas (set_values, MFP_thermally_perfect_gas.cpp:24-26)."""
    v = gas[key]
    if isinstance(v, (list, tuple)):
        if len(v) != (gas_n_tracers(gas) + 1):
            raise ValueError("'mass', 'charge' and 'gamma' must have the same number of components")  # :37-40
        if len(v) - 1 > abi.MAX_TRACERS:
            raise NotImplementedError(f"at most {abi.MAX_TRACERS} tracers")
        return [float(x) for x in v]
    return [float(v)] * (gas_n_tracers(gas) + 1)


def gas_is_mixture(gas):
    """True where the species differ in mass, charge or gamma (mass_const / charge_const / gamma_const false,
    MFP_thermally_perfect_gas.cpp:32-34): gamma, mass, charge and cp of a cell then follow its mass fractions."""
    return any(x != v[0] for v in (gas_list(gas, "mass"), gas_list(gas, "gamma"), gas_list(gas, "charge")) for x in v)


def gas_comp_names(gas, state_name):
    """Names of the sub-components of a gas: gas.names, else the state's name (one species) or <state>_<i>
    (ThermallyPerfectGas, MFP_thermally_perfect_gas.cpp:27,42-48)."""
    names = gas.get("names")
    if names:
        return [names] if isinstance(names, str) else list(names)
    n = gas_n_tracers(gas) + 1
    return [state_name] if n == 1 else [f"{state_name}_{i}" for i in range(n)]


def gas_scalar(gas, key):
    """mass / charge / gamma of species 0: the constants of the gas where every species has the same
    (mass_const, charge_const, gamma_const, MFP_thermally_perfect_gas.cpp:33-35: the tracers are then
    passive scalars)."""
    return gas_list(gas, key)[0]


def gas_gamma_from_alpha(gas, alpha):
    """ThermallyPerfectGas::get_gamma_from_prim (MFP_thermally_perfect_gas.cpp:51-83) on arrays:
    alpha = list of the n_tracers mass-fraction arrays."""
    g, m = gas_list(gas, "gamma"), gas_list(gas, "mass")
    if all(x == g[0] for x in g):
        return g[0]
    s_cp = s_cv = s_a = 0.0
    for i, a in enumerate(alpha):
        a = np.clip(a, 0.0, 1.0)
        s_a = s_a + a
        s_cp = s_cp + a * (g[i] / (m[i] * (g[i] - 1.0)))
        s_cv = s_cv + a * (1.0 / (m[i] * (g[i] - 1.0)))
    s_cp = s_cp + (1.0 - s_a) * (g[-1] / (m[-1] * (g[-1] - 1.0)))
    s_cv = s_cv + (1.0 - s_a) * (1.0 / (m[-1] * (g[-1] - 1.0)))
    return s_cp / s_cv


def gas_cp_from_alpha(gas, alpha):
    """ThermallyPerfectGas::get_cp_from_prim (MFP_thermally_perfect_gas.cpp:123-147) on arrays."""
    g, m = gas_list(gas, "gamma"), gas_list(gas, "mass")
    if all(x == g[0] for x in g) and all(x == m[0] for x in m):
        return g[0] / (m[0] * (g[0] - 1))
    s_cp = s_a = 0.0
    for i, a in enumerate(alpha):
        a = np.clip(a, 0.0, 1.0)
        s_a = s_a + a
        s_cp = s_cp + a * (g[i] / (m[i] * (g[i] - 1.0)))
    return s_cp + (1.0 - s_a) * (g[-1] / (m[-1] * (g[-1] - 1.0)))


def gas_charge_from_alpha(gas, alpha):
    """HydroGas::get_charge_from_prim (MFP_hydro_gas.cpp:98-121) on arrays."""
    q, m = gas_list(gas, "charge"), gas_list(gas, "mass")
    if all(x == q[0] for x in q):
        return q[0]
    s_q = s_m = s_a = 0.0
    for i, a in enumerate(alpha):
        a = np.clip(a, 0.0, 1.0)
        s_q = s_q + a * q[i] / m[i]
        s_m = s_m + a / m[i]
        s_a = s_a + a
    s_q = s_q + (1.0 - s_a) * q[-1] / m[-1]
    s_m = s_m + (1.0 - s_a) / m[-1]
    return s_q / s_m


def gas_mass_from_alpha(gas, alpha):
    """HydroGas::get_mass_from_prim (MFP_hydro_gas.cpp:50-70) on arrays."""
    m = gas_list(gas, "mass")
    if all(x == m[0] for x in m):
        return m[0]
    s_m = s_a = 0.0
    for i, a in enumerate(alpha):
        a = np.clip(a, 0.0, 1.0)
        s_a = s_a + a
        s_m = s_m + a / m[i]
    s_m = s_m + (1.0 - s_a) / m[-1]
    return 1.0 / s_m


class Problem:
    """Sorted states/actions and everything cmfp_create needs."""

    def __init__(self, cfg, fp_mode=abi.FP_STRICT):
        self.cfg = cfg
        self.dim = int(cfg["dim"])
        self.n_cell = [int(n) for n in cfg["n_cell"]] + [1] * (3 - self.dim)
        self.prob_lo = [float(v) for v in cfg["prob_lo"]] + [0.0] * (3 - self.dim)
        self.prob_hi = [float(v) for v in cfg["prob_hi"]] + [1.0] * (3 - self.dim)
        self.periodic = [int(v) for v in cfg["is_periodic"]] + [0] * (3 - self.dim)
        # Geometry::CellSize: (prob_hi - prob_lo) / n_cell
        self.dx = [(self.prob_hi[d] - self.prob_lo[d]) / self.n_cell[d] for d in range(3)]
        self.ref = update_ref(cfg)
        self.cfl = float(cfg.get("cfl", DEFAULTS["cfl"]))
        self.scheme = cfg.get("time_integration_scheme", DEFAULTS["time_integration_scheme"])
        self.force_dt = float(cfg.get("force_dt", DEFAULTS["force_dt"]))   # MFP_timestep.cpp:8
        self.fp_mode = fp_mode
        self.state_names = sorted(cfg["states"].keys())       # get_sorted_keys, MFP_config.cpp:102
        # `elliptic` actions act through apply_change only (MFP_advance.cpp:49): they are kept beside the actions
        # of the hyperbolic update, in the same sorted order
        acts = cfg.get("actions", {})
        self.action_names = sorted(k for k in acts if acts[k]["type"] != "elliptic")  # :261
        self.elliptic_names = sorted(k for k in acts if acts[k]["type"] == "elliptic")
        self.state_index = {n: i for i, n in enumerate(self.state_names)}

    # -- descriptors -----------------------------------------------------
    def geometry(self, block_lo=None, block_n=None):
        g = abi.Geometry()
        g.dim = self.dim
        for d in range(3):
            g.n_cell[d] = self.n_cell[d]
            g.prob_lo[d] = self.prob_lo[d]
            g.dx[d] = self.dx[d]
            g.is_periodic[d] = self.periodic[d]
            g.block_lo[d] = 0 if block_lo is None else block_lo[d]
            g.block_n[d] = self.n_cell[d] if block_n is None else block_n[d]
        return g

    def params(self):
        p = abi.Params()
        p.lightspeed = self.ref["lightspeed"]
        p.Larmor = self.ref["Larmor"]
        p.Debye = self.ref["Debye"]
        p.cfl = self.cfl
        p.time_integrator = abi.TIME_INTEGRATOR[self.scheme]
        p.fp_mode = self.fp_mode
        p.n0 = self.ref["n0"]
        p.force_dt = self.force_dt
        return p

    def state_descs(self):
        arr = (abi.StateDesc * len(self.state_names))()
        for i, name in enumerate(self.state_names):
            sd = self.cfg["states"][name]
            d = arr[i]
            d.reconstruction = abi.RECON[sd["reconstruction"]]
            d.shock_threshold = -1.0
            d.effective_zero = float(sd.get("effective_zero", 1e-14))   # MFP_eulerian.cpp:62
            if sd["type"] == "hydro":
                d.kind = abi.STATE_HYDRO
                gas = sd["gas"]
                if gas.get("type", "thermally_perfect") != "thermally_perfect":
                    raise NotImplementedError("only the thermally_perfect gas is on the built path")
                d.mass, d.charge, d.gamma = (gas_scalar(gas, k) for k in ("mass", "charge", "gamma"))
                d.n_tracers = gas_n_tracers(gas)
                if gas_is_mixture(gas):
                    d.mixture = 1
                    for k, x in enumerate(gas_list(gas, "mass")):
                        d.species_mass[k] = x
                    for k, x in enumerate(gas_list(gas, "gamma")):
                        d.species_gamma[k] = x
                    for k, x in enumerate(gas_list(gas, "charge")):
                        d.species_charge[k] = x
                d.flux = abi.FLUX[sd["flux"]]
                if d.flux == abi.FLUX["RankineHugoniot"]:
                    raise ValueError("hydro state with a field flux")
                if "shock_detector" in sd:
                    if sd["shock_detector"].get("name") != "pressure_jump_detector":
                        raise ValueError("unknown shock detector")
                    d.shock_threshold = float(sd["shock_detector"]["threshold"])
                visc = sd.get("viscosity")
                if visc:                                   # MFP_hydro.cpp:86-111, MFP_hydro_viscous.cpp
                    if visc["type"] not in abi.VISCOSITY:
                        raise ValueError(f"Invalid viscosity option '{visc['type']}'")
                    d.viscosity = abi.VISCOSITY[visc["type"]]
                    r = self.ref
                    d.visc_Pr = float(visc["Pr"])
                    if visc["type"] == "UserDefined":
                        d.visc_mu0 = float(visc["mu0"])            # :257-258: taken as given
                        ok = d.visc_mu0 > 0
                    else:
                        d.visc_mu0 = float(visc["mu0"]) / (r["rho_ref"] * r["x_ref"] * r["u_ref"])
                        d.visc_T0 = float(visc["T0"]) / r["T_ref"]
                        if visc["type"] == "Sutherland":
                            d.visc_S = float(visc["S"]) / r["T_ref"]
                            ok = d.visc_mu0 > 0 and d.visc_T0 > 0 and d.visc_S > 0
                        else:
                            d.visc_n = float(visc["n"])
                            ok = d.visc_mu0 > 0 and d.visc_T0 > 0 and d.visc_n > 0
                    if not ok:
                        raise ValueError(f"{visc['type']} input coefficients non-physical")
                table = bc.hydro_bc(sd.get("bc", {}), self.dim, self.periodic, d.n_tracers)
            elif sd["type"] == "field":
                d.kind = abi.STATE_FIELD
                d.mass, d.charge, d.gamma = 1.0, 0.0, 1.0
                if sd["flux"] != "RankineHugoniot":
                    raise NotImplementedError("field 'HLLE' reads uninitialised memory in the reference "
                                              "(SURVEY quirk Q4); only RankineHugoniot is built")
                d.flux = abi.FLUX[sd["flux"]]
                d.div_transport = float(sd.get("div_transport", 0.0))
                d.div_damping = float(sd.get("div_damping", 0.0))
                table = bc.field_bc(sd.get("bc", {}), self.dim, self.periodic)
            else:
                raise NotImplementedError(f"state type '{sd['type']}' is outside the built path")
            for c in range(abi.MAX_COMP):
                for dd in range(abi.MAX_DIM):
                    for s in range(2):
                        d.bc[c][dd][s] = table[c][dd][s]
        return arr

    def action_descs(self):
        arr = (abi.ActionDesc * max(1, len(self.action_names)))()
        for i, name in enumerate(self.action_names):
            ad = self.cfg["actions"][name]
            a = arr[i]
            a.kind = abi.ACTION[ad["type"]]
            a.corner_transport = int(bool(ad.get("corner_transport", True)))
            a.solver = abi.SOLVER[ad.get("solver", "explicit")]
            names = list(ad["states"])
            idx = [self.state_index[n] for n in names]
            if ad["type"] in ("plasma5", "Lorentz"):
                kinds = [self.cfg["states"][n]["type"] for n in names]
                species = [i2 for i2, k in zip(idx, kinds) if k == "hydro"]
                field = [i2 for i2, k in zip(idx, kinds) if k == "field"]
                if len(field) != 1:
                    raise ValueError(f"action '{name}' needs exactly one field state")
                idx = species + field       # species in list order, field last (MFP_plasma5.cpp:35-55)
            if ad["type"] == "collisions_rambo":
                if any(self.cfg["states"][n]["type"] != "hydro" for n in names):
                    raise ValueError(f"An invalid state has been defined for the Collisions source {name}")
                # the species of the action: 'species' (names of gas sub-components), else the states themselves
                # (init_species_info MFP_collisions.cpp:65-83); each is looked up in the states' gas.names (:43-60)
                species = list(ad.get("species") or names)
                where = []
                for sn in species:
                    hit = None
                    for si, stn in enumerate(names):
                        comps = gas_comp_names(self.cfg["states"][stn]["gas"], stn)
                        if sn in comps:
                            hit = (si, comps.index(sn))
                            break
                    if hit is None:
                        raise ValueError(f"Couldn't find sub-component for '{sn}'")
                    where.append(hit)
                # the reference reads U, Q, alpha and the gas of species n from entry n of its state list
                # (MFP_collisions.cpp:181-190): defined only where species n lives in state n
                if [w[0] for w in where] != list(range(len(names))):
                    raise NotImplementedError(
                        f"action '{name}': species n must be a sub-component of state n of the 'states' list (the reference "
                        "indexes the states' data with the species index, MFP_collisions.cpp:181-190)")
                for n, (si, k) in enumerate(where):
                    a.species_component[n] = k
                charges = [gas_list(self.cfg["states"][names[si]]["gas"], "charge")[k] for si, k in where]
                # cross_sections = {a = {b = sigma}}: non-dimensionalised and cross-loaded
                # (MFP_collisions_rambo.cpp:39-55); names that are not species of this action are skipped
                for na, row in ad.get("cross_sections", {}).items():
                    if na not in species:
                        continue
                    for nb, sigma in row.items():
                        if nb not in species:
                            continue
                        ia, ib = species.index(na), species.index(nb)
                        s = float(sigma) / (self.ref["x_ref"] * self.ref["x_ref"])
                        a.cross_sections[ia][ib] = s
                        a.cross_sections[ib][ia] = s
                if not ad.get("cross_sections"):
                    for sn, q in zip(species, charges):
                        if q == 0.0:
                            raise ValueError(f"cross sections for collisions involving neutral species '{sn}' are "
                                             f"required by action '{name}'")
            a.n_states = len(idx)
            for j, v in enumerate(idx):
                a.states[j] = v
        return arr

    def elliptic_actions(self):
        """[(field state index, [species state indices])] of the `elliptic` actions, sorted by name:
        Elliptic::Elliptic (MFP_elliptic.cpp:31-68) and the species of the plasma5 actions associated with the
        field (:722-755).  Only projection = 1 is built (the static-field solve is outside the path)."""
        out = []
        for name in self.elliptic_names:
            ad = self.cfg["actions"][name]
            if int(ad.get("projection", 0)) != 1:
                raise NotImplementedError(f"action '{name}': only projection = 1 of 'elliptic' is built")
            st = ad.get("state", "null")
            if st == "null":
                raise ValueError(f"Action '{name}' requires option 'state' to be defined.")
            if self.cfg["states"][st]["type"] != "field":
                raise ValueError(f"An invalid state has been defined for the '{name}' action")
            # project_divergence_field zeroes the charge density inside its loop over the field's associated
            # actions (`local_cd.setVal(0.0)`, MFP_elliptic.cpp:722-735) and divides by ep per action: with several
            # plasma5 actions on one field only the last one, in sorted-name order, contributes.  Reproduced.
            species = []
            for an in self.action_names:
                a = self.cfg["actions"][an]
                if a["type"] == "plasma5" and st in a["states"]:
                    species = [self.state_index[n] for n in a["states"] if self.cfg["states"][n]["type"] == "hydro"]
            out.append((self.state_index[st], species))
        return out

    def elliptic_face_types(self, field_state):
        """(bc_B, bc_D): the physical boundary type (FieldState::bc_names, MFP_field.cpp:36-40) of the field's
        fill_B_bc / fill_D_bc entry per face [2 * dim + side] - what Elliptic's set_field_bcs (MFP_elliptic.cpp:83-122)
        reads from phys_fill_bc[Bx] / phys_fill_bc[Dx]; 'interior' (0) in periodic directions (MFP_field.cpp:415-420)."""
        from . import bc as bcmod
        sd = self.cfg["states"][self.state_names[field_state]]
        out = []
        for key in ("fill_B_bc", "fill_D_bc"):
            t = [0] * 6
            for d in range(self.dim):
                for side in range(2):
                    name = bcmod._get(sd.get("bc", {}), d, side, key, "outflow")
                    t[2 * d + side] = 0 if self.periodic[d] else bcmod.FIELD_BC_NAMES[name]
            out.append(t)
        return tuple(out)

    def n_comp(self, s):
        sd = self.cfg["states"][self.state_names[s]]
        return 5 + gas_n_tracers(sd["gas"]) if sd["type"] == "hydro" else 10

    # -- Dirichlet data of the domain faces ----------------------------------
    def face_entries(self, s):
        sd = self.cfg["states"][self.state_names[s]]
        nt = gas_n_tracers(sd["gas"]) if sd["type"] == "hydro" else 0
        return bc.face_values(sd, self.dim, self.periodic, nt)

    def face_tables(self, s, ng, t, block_lo=None, block_n=None):
        """Tabulated values of the per-face numbers / functions of (x, y, z, t) of state s:
        [(d, side, prim, cell_values, face_values)] over the boxes cmfp_state_set_face_values
        describes (block-local, ghost width ng).  Cell tables at the ghost-cell centres
        (update_boundary_cells, MFP_eulerian.cpp:478-489), face tables at the face centres (offset 0
        along the face normal, face_bc :841-855)."""
        lo = [0, 0, 0] if block_lo is None else list(block_lo)
        n = list(self.n_cell) if block_n is None else list(block_n)

        def positions(d, side, face):
            ax = []
            for e in range(3):
                if e >= self.dim:
                    idx, off = np.zeros(1), 0.5
                elif e != d:
                    idx, off = np.arange(lo[e] - ng, lo[e] + n[e] + ng, dtype=np.float64), 0.5
                elif face:
                    idx, off = np.array([lo[e] if side == 0 else lo[e] + n[e]], dtype=np.float64), 0.0
                else:
                    a0 = lo[e] - ng if side == 0 else lo[e] + n[e]
                    idx, off = np.arange(a0, a0 + ng, dtype=np.float64), 0.5
                ax.append(self.prob_lo[e] + (idx + off) * self.dx[e])
            z, y, x = np.meshgrid(ax[2], ax[1], ax[0], indexing="ij")
            return {"x": x, "y": y, "z": z, "t": t}

        out = []
        for (d, side), vals in self.face_entries(s).items():
            # every block registers every entry, also for faces it does not touch (the library applies a
            # table only on a physical face of its block): the choice of kernel path then is the same
            # on all ranks, which the halo exchange pattern relies on
            cdat, fdat = positions(d, side, False), positions(d, side, True)
            for prim, v in vals.items():
                cell = np.ascontiguousarray(self._eval(v, cdat, cdat["x"].shape))
                face = np.ascontiguousarray(self._eval(v, fdat, fdat["x"].shape))
                out.append((d, side, prim, cell, face))
        return out

    def face_values_depend_on_time(self, s):
        return any(getattr(v, "uses_time", False) for vals in self.face_entries(s).values() for v in vals.values())

    # -- 'dynamic' components ------------------------------------------------
    def dynamic_entries(self, s):
        """{conserved component: its `value` entry} of states.<name>.dynamic = {'x_B', ...}: FieldState::set_udf
        (MFP_field.cpp:175-221) points dynamic_functions[i] at the initialisation function of every listed
        component.  HydroState::set_udf does not read the key: ignored there, as in the reference."""
        sd = self.cfg["states"][self.state_names[s]]
        if sd["type"] != "field":
            return {}
        names = sd.get("dynamic") or []
        names = [names] if isinstance(names, str) else list(names)
        val = sd.get("value", {})
        return {c: val.get(n, FIELD_DEFAULT.get(n, 0.0)) for c, n in enumerate(FIELD_CONS) if n in names}

    def dynamic_depends_on_time(self, s):
        return any(getattr(v, "uses_time", False) for v in self.dynamic_entries(s).values())

    def dynamic_tables(self, s, ng, t, block_lo=None, block_n=None):
        """[(component, values)]: the functions at the centres of the block's cells and of ng ghost layers around
        it (calc_primitives runs over grow(box, num_grow): x = prob_lo + (i + 0.5) dx, MFP_eulerian.cpp:349-355)."""
        lo = [0, 0, 0] if block_lo is None else list(block_lo)
        n = list(self.n_cell) if block_n is None else list(block_n)
        glo = [lo[d] - ng if d < self.dim else 0 for d in range(3)]
        gn = [n[d] + 2 * ng if d < self.dim else 1 for d in range(3)]
        dat = self.cell_centres(glo, gn)
        dat["t"] = t
        return [(c, np.ascontiguousarray(self._eval(v, dat, dat["x"].shape))) for c, v in self.dynamic_entries(s).items()]

    # -- initial data ----------------------------------------------------
    def cell_centres(self, lo=None, n=None):
        lo = [0, 0, 0] if lo is None else list(lo) + [0] * (3 - len(lo))
        n = self.n_cell if n is None else list(n) + [1] * (3 - len(n))
        ax = []
        for d in range(3):
            i = np.arange(lo[d], lo[d] + n[d], dtype=np.float64)
            ax.append(self.prob_lo[d] + (i + 0.5) * self.dx[d])   # MFP_hydro.cpp:531-536
        z, y, x = np.meshgrid(ax[2], ax[1], ax[0], indexing="ij")
        return {"x": x, "y": y, "z": z, "t": 0.0}

    @staticmethod
    def _eval(v, dat, shape, default=0.0):
        if v is None:
            return np.full(shape, default)
        if callable(v):
            return np.broadcast_to(np.asarray(v(dat), dtype=np.float64), shape).copy()
        return np.full(shape, float(v))

    def initial_data(self, s, lo=None, n=None):
        """Conserved initial data [comp][k][j][i] of state s on a block:
        HydroState::init_data (MFP_hydro.cpp:504-567) with define_rho_p_T and
        prim2cons (MFP_thermally_perfect_gas.cpp:248-269,220-246);
        EulerianState::init_data for the field (MFP_eulerian.cpp:80-139)."""
        sd = self.cfg["states"][self.state_names[s]]
        dat = self.cell_centres(lo, n)
        shape = dat["x"].shape
        val = sd.get("value", {})
        if sd["type"] == "field":
            out = np.empty((10,) + shape)
            for c, name in enumerate(FIELD_CONS):
                out[c] = self._eval(val.get(name), dat, shape, FIELD_DEFAULT.get(name, 0.0))
            return out
        gas = sd["gas"]
        nt = gas_n_tracers(gas)
        alpha = val.get("alpha", [])
        if not isinstance(alpha, (list, tuple)):
            alpha = [alpha] * nt
        alpha = [self._eval(alpha[i] if i < len(alpha) else None, dat, shape) for i in range(nt)]
        m, g = gas_mass_from_alpha(gas, alpha), gas_gamma_from_alpha(gas, alpha)   # get_mass_from_prim, get_gamma_from_prim
        rho = self._eval(val.get("rho"), dat, shape) if "rho" in val else None
        if rho is None and "nd" in val:                       # init_from_number_density, MFP_hydro.cpp:113-120
            rho = self._eval(val["nd"], dat, shape) * m
        if rho is None:
            rho = np.zeros(shape)
        u = self._eval(val.get("x_vel"), dat, shape)
        v = self._eval(val.get("y_vel"), dat, shape)
        w = self._eval(val.get("z_vel"), dat, shape)
        p = self._eval(val.get("p"), dat, shape)
        T = self._eval(val.get("T"), dat, shape)
        # define_rho_p_T: fill the missing one of rho, p, T
        a = (rho > 0) & (p > 0)
        b = ~a & (p > 0) & (T > 0)
        c = ~a & ~b & (rho > 0) & (T > 0)
        rho = np.where(b, p * m / np.where(T != 0, T, 1.0), rho)
        p = np.where(c, rho * T / m, p)
        if np.any(rho <= 0) or np.any(p <= 0):
            raise ValueError("Primitive values outside of physical bounds!!")
        out = np.empty((5 + nt,) + shape)
        for i in range(nt):                                   # tracer = alpha * rho, MFP_thermally_perfect_gas.cpp:243-245
            out[5 + i] = alpha[i] * rho
        out[0] = rho
        out[1] = u * rho
        out[2] = v * rho
        out[3] = w * rho
        ke = 0.5 * rho * (u * u + v * v + w * w)
        out[4] = p / (g - 1) + ke
        return out
