"""ctypes front-end of the CPU oracle (oracle/libmfp_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Never imported by cerberus_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

dp = C.POINTER(C.c_double)


def build():
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libmfp_oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_create.restype = C.c_void_p
        L.orc_est_time_step.restype = C.c_double
        L.orc_state_allowed_dt.restype = C.c_double
        L.orc_action_allowed_dt.restype = C.c_double
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(dp)


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


# ---- leaves (SoA [comp][n]) -------------------------------------------------
def reconstruct(recon, stencil):
    stencil = _c(stencil)
    n = stencil.shape[1]
    slope, lo, hi = np.empty(n), np.empty(n), np.empty(n)
    lib().orc_reconstruct(C.c_int(recon), C.c_long(n), _p(stencil), _p(slope), _p(lo), _p(hi))
    return slope, lo, hi


def riemann(flux, L, R, c0=0.0, ch=0.0, shock_threshold=-1.0):
    L, R = _c(L), _c(R)
    n = L.shape[1]
    F = np.empty((10 if flux == 10 else 5, n))
    lib().orc_riemann(C.c_int(flux), C.c_long(n), _p(L), _p(R), _p(F), C.c_double(c0), C.c_double(ch),
                      C.c_double(shock_threshold))
    return F


def cons2prim(mass, gamma, U):
    U = _c(U)
    n = U.shape[1]
    Q, s = np.empty((8, n)), np.empty((3, n))
    lib().orc_cons2prim(C.c_long(n), C.c_double(mass), C.c_double(gamma), _p(U), _p(Q), _p(s))
    return Q, s


def prim2cons(mass, gamma, Q):
    Q = _c(Q)
    n = Q.shape[1]
    U = np.empty((5, n))
    lib().orc_prim2cons(C.c_long(n), C.c_double(mass), C.c_double(gamma), _p(Q), _p(U))
    return U


def colpiv_solve(A, b):
    A, b = _c(A), _c(b)
    x = np.empty_like(b)
    lib().orc_colpiv_solve(C.c_int(len(b)), _p(A), _p(b), _p(x))
    return x


# ---- divergence projection (numpy restatement) -------------------------------
def _node_ops(dim):
    """(difference, average) along one axis between cells and nodes on a periodic grid.  Arrays are
    [k][j][i]; node (i, j, k) is the lower corner of cell (i, j, k)."""
    ax = {0: -1, 1: -2, 2: -3}

    def to_nodes(v, d, diff):      # cells i-1, i -> node i
        m = np.roll(v, 1, axis=ax[d])
        return v - m if diff else v + m

    def to_cells(p, d, diff):      # nodes i, i+1 -> cell i
        q = np.roll(p, -1, axis=ax[d])
        return q - p if diff else q + p
    return to_nodes, to_cells


def solve_divergence(V, dx, dim, S_cc=None):
    """Elliptic::solve_divergence (MFP_elliptic.cpp:784-964) with AMReX's nodal operators [AMReX absent: restated
    from its published discretisation, see cerberus_amd/csrc/mfp_elliptic.hip] on a periodic grid, solved by FFT.
    V: [dim][k][j][i] cell-centred vector.  Returns (V_new, phi at cell centres)."""
    to_nodes, to_cells = _node_ops(dim)
    fac = 0.5 ** (dim - 1)
    rhs = np.zeros(V.shape[1:])
    for d in range(dim):
        t = V[d]
        for e in range(dim):
            t = to_nodes(t, e, e == d)
        rhs = rhs + (fac / dx[d]) * t
    if S_cc is not None:
        t = S_cc
        for e in range(dim):
            t = to_nodes(t, e, False)
        rhs = rhs + t * (0.5 ** dim)
    n = [V.shape[-1], V.shape[-2], V.shape[-3]]
    th = [2 * np.pi * np.fft.fftfreq(n[d]) for d in range(dim)]
    K = [(2 - 2 * np.cos(th[d])) / dx[d] ** 2 for d in range(dim)]
    M = [(4 + 2 * np.cos(th[d])) / 6 for d in range(dim)]
    shape = lambda a, d: a.reshape([-1 if (2 - e) == d else 1 for e in range(3)])
    sym = np.zeros([n[2] if dim == 3 else 1, n[1], n[0]])
    for d in range(dim):
        term = shape(K[d], d)
        for e in range(dim):
            if e != d:
                term = term * shape(M[e], e)
        sym = sym - term
    axes = (-1, -2, -3)[:dim]
    hat = np.fft.fftn(rhs, axes=axes)
    sym[0, 0, 0] = 1.0
    hat = hat / sym
    hat[0, 0, 0] = 0.0
    phi = np.real(np.fft.ifftn(hat, axes=axes))
    Vn = V.copy()
    for d in range(dim):
        t = phi
        for e in range(dim):
            t = to_cells(t, e, e == d)
        Vn[d] = V[d] - (fac / dx[d]) * t
    t = phi
    for e in range(dim):
        t = to_cells(t, e, False)
    return Vn, t * (0.5 ** dim)


def project_divergence_field(F, rhos, rs, dx, dim, cd_scale):
    """Elliptic::project_divergence_field (MFP_elliptic.cpp:667-782) on the field components F [10][k][j][i]."""
    F = F.copy()
    Vb = np.stack([F[3], F[4], F[5]])
    Vb[:dim], F[7] = solve_divergence(Vb[:dim], dx, dim)
    F[3:3 + dim] = Vb[:dim]
    cd = np.zeros_like(F[0])
    for rho, r in zip(rhos, rs):
        cd = cd + rho * r
    S = (cd / F[9]) * (-cd_scale)
    Vd = np.stack([F[0], F[1], F[2]])
    Vd[:dim], F[6] = solve_divergence(Vd[:dim], dx, dim, S)
    F[0:dim] = Vd[:dim]
    return F


# ---- level ------------------------------------------------------------------
class OracleSim:
    """Whole-level CPU restatement driven from a cerberus_amd.config.Problem."""

    def __init__(self, problem, max_grid_size=0):
        self.p = problem
        L = lib()
        g, pr = problem.geometry(), problem.params()
        sd, ad = problem.state_descs(), problem.action_descs()
        self._keep = (g, pr, sd, ad)
        self.h = C.c_void_p(L.orc_create(C.byref(g), C.byref(pr), C.c_int(len(problem.state_names)), sd,
                                         C.c_int(len(problem.action_names)), ad, C.c_int(max_grid_size)))
        self.shape = tuple(problem.n_cell[::-1])
        for s in range(len(problem.state_names)):
            self.set_state(s, problem.initial_data(s))
        self.time = 0.0
        self._time_bc = [s for s in range(len(problem.state_names)) if problem.face_values_depend_on_time(s)]
        for s in range(len(problem.state_names)):
            self._set_face_values(s, 0.0)

    TNG = 4    # ghost width of the Dirichlet tables handed to the oracle (>= every num_grow)

    def _set_face_values(self, s, t):
        sd = self.p.cfg["states"][self.p.state_names[s]]
        for d, side, prim, cell, face in self.p.face_tables(s, self.TNG, t):
            # carried primitive index -> index in the reference's primitive vector (alpha behind the 8 hydro prims)
            rp = prim if (sd["type"] != "hydro" or prim < 5) else prim + 3
            lib().orc_set_face_values(self.h, C.c_int(s), C.c_int(d), C.c_int(side), C.c_int(rp), C.c_int(self.TNG),
                                      _p(cell), _p(face))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_destroy(self.h)
            self.h = None

    def set_state(self, s, data):
        data = _c(data)
        assert data.shape == (self.p.n_comp(s),) + self.shape, data.shape
        lib().orc_set_state(self.h, C.c_int(s), _p(data))

    def get_state(self, s):
        out = np.empty((self.p.n_comp(s),) + self.shape)
        lib().orc_get_state(self.h, C.c_int(s), _p(out))
        return out

    def advance(self, dt):
        for s in self._time_bc:
            self._set_face_values(s, self.time + dt)
        self.time += dt
        lib().orc_clear_error()
        lib().orc_advance(self.h, C.c_double(dt))
        if lib().orc_error_flag():
            raise FloatingPointError("Primitive values outside of physical bounds!!")
        for fs, species in self.p.elliptic_actions():      # act->apply_change (MFP_advance.cpp:49)
            if not all(self.p.periodic[:self.p.dim]):
                raise NotImplementedError("divergence projection: periodic dimensions only")
            r = self.p.ref
            gas = lambda s: self.p.cfg["states"][self.p.state_names[s]]["gas"]

            def r_of(s):     # get_charge_from_cons / get_mass_from_cons (HydroState::calc_current_and_charge, MFP_hydro.cpp:1720-1721)
                from cerberus_amd import config
                U = self.get_state(s)
                al = [U[5 + i] / U[0] for i in range(config.gas_n_tracers(gas(s)))]
                return config.gas_charge_from_alpha(gas(s), al) / config.gas_mass_from_alpha(gas(s), al)
            F = project_divergence_field(self.get_state(fs), [self.get_state(s)[0] for s in species],
                                         [r_of(s) for s in species],
                                         self.p.dx, self.p.dim, r["Larmor"] / (r["Debye"] ** 2 * r["lightspeed"]))
            self.set_state(fs, F)

    def est_time_step(self):
        return lib().orc_est_time_step(self.h)

    def state_allowed_dt(self, s):
        return lib().orc_state_allowed_dt(self.h, C.c_int(s))

    def action_allowed_dt(self, a):
        return lib().orc_action_allowed_dt(self.h, C.c_int(a))

    def action_dU(self, a, dt, s):
        out = np.zeros((self.p.n_comp(s),) + self.shape)
        lib().orc_action_dU(self.h, C.c_int(a), C.c_double(dt), C.c_int(s), _p(out))
        return out

    def fill_patch(self, s):
        ng = abi_ngrow(self.p, s)
        shp = tuple((n + 2 * ng) if d < self.p.dim else n for d, n in enumerate(self.p.n_cell))[::-1]
        out = np.empty((self.p.n_comp(s),) + shp)
        lib().orc_fill_patch_global(self.h, C.c_int(s), _p(out))
        return out

    def run(self, nsteps, dt0=None):
        """Amr::coarseTimeStep loop: computeInitialDt then computeNewDt's 1.1x
        growth cap (MFP_timestep.cpp:23-119), no stop_time."""
        dts = []
        dt = self.est_time_step() if dt0 is None else dt0
        for _ in range(nsteps):
            self.advance(dt)
            dts.append(dt)
            dt = min(self.est_time_step(), 1.1 * dt)
        return dts


def abi_ngrow(problem, s):
    from cerberus_amd import abi
    name = problem.state_names[s]
    return abi.RECON_NGROW[abi.RECON[problem.cfg["states"][name]["reconstruction"]]]
