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


# ---- the same projection on a domain with physical faces ------------------------------------------------------
# LinOpBCType of a face as Elliptic's set_field_bcs picks it (MFP_elliptic.cpp:83-122): periodic direction ->
# Periodic; outflow, symmetry -> Neumann; inflow -> Dirichlet (homogeneous).  'asymmetry' maps to
# LinOpBCType::reflect_odd there (the field's bc_names give it the value of PhysBCType::slipwall,
# MFP_field.cpp:36-40), which AMReX's nodal solver does not define: refused.
PERIODIC, NEUMANN, DIRICHLET = 0, 1, 2
LINOP_OF_PHYS = {"interior": None, "inflow": DIRICHLET, "outflow": NEUMANN, "symmetry": NEUMANN, "asymmetry": None}


def _ops_1d(n, h, lo, hi):
    """stiffness and mass matrices of the linear elements along one direction as the nodal Laplacian applies them:
    periodic (n nodes), or nodes 0 .. n with the ghost node of a Neumann face mirrored (row doubled) and the node
    of a Dirichlet face removed.  Returns (K, M, free node indices, weights of the mirrored extension)."""
    import scipy.sparse as sp
    if lo == PERIODIC:
        e = np.ones(n)
        K = sp.diags([2 * e, -e[1:], -e[1:]], [0, 1, -1], format="lil")
        M = sp.diags([4 * e, e[1:], e[1:]], [0, 1, -1], format="lil")
        if n > 1:
            K[0, n - 1] += -1
            K[n - 1, 0] += -1
            M[0, n - 1] += 1
            M[n - 1, 0] += 1
        return (K / h ** 2).tocsr(), (M / 6.0).tocsr(), np.arange(n), np.ones(n)
    m = n + 1
    e = np.ones(m)
    K = sp.diags([2 * e, -e[1:], -e[1:]], [0, 1, -1], format="lil")
    M = sp.diags([4 * e, e[1:], e[1:]], [0, 1, -1], format="lil")
    w = np.ones(m)
    if lo == NEUMANN:
        K[0, 1] = -2
        M[0, 1] = 2
        w[0] = 0.5
    if hi == NEUMANN:
        K[n, n - 1] = -2
        M[n, n - 1] = 2
        w[n] = 0.5
    free = np.arange(m)
    if lo == DIRICHLET:
        free = free[free != 0]
    if hi == DIRICHLET:
        free = free[free != n]
    return (K / h ** 2).tocsr(), (M / 6.0).tocsr(), free, w


def solve_divergence_bc(Vg, dx, dim, bc, S_cc=None):
    """Elliptic::solve_divergence (MFP_elliptic.cpp:784-964) on a domain with physical faces.  Vg: the cell-centred
    vector [dim][k][j][i] WITH ONE GHOST LAYER in every used dimension, filled as FillPatch fills it (:897-904);
    bc[d] = (lo, hi) of PERIODIC / NEUMANN / DIRICHLET; S_cc on the valid cells (zero outside, :704-714).
    AMReX's nodal operators, restated from memory of its published discretisation [AMReX absent: PARITY UNPINNED]:
      rhs (compRHS): nodal divergence of the cells around a node; at a node on a Neumann face the cells beyond that
        face contribute their normal component only (mlndlap_divu zeroes the tangential ones) and no source, then
        the node's value is doubled per Neumann face it lies on (mlndlap_impose_neumann_bc); 0 at Dirichlet nodes;
      operator: the finite-element stencil with the ghost nodes of a Neumann face mirrored, phi = 0 on Dirichlet
        faces; without a Dirichlet face the constant is fixed by a zero mean over the mirrored extension.
    Solved here by assembling the sparse operator and a direct solve - no transform, so that the HIP path
    (mirrored extension + FFT) is checked against an independent method.  Returns (V_new valid cells, phi_cc)."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl
    n = [Vg.shape[-1 - d] - 2 for d in range(dim)]
    per = [bc[d][0] == PERIODIC for d in range(dim)]
    nn = [n[d] if per[d] else n[d] + 1 for d in range(dim)]
    shape = [1, 1, 1]
    for d in range(dim):
        shape[2 - d] = nn[d]
    fac = 0.5 ** (dim - 1)
    rhs = np.zeros(shape)
    node = [np.arange(nn[d]) for d in range(dim)]
    for corner in range(1 << dim):
        a = [(corner >> d) & 1 for d in range(dim)]
        idx, inside = [], []
        for d in range(dim):
            c = node[d] - 1 + a[d]
            if per[d]:
                idx.append(c % n[d] + 1)
                inside.append(np.ones(nn[d], dtype=bool))
            else:
                idx.append(c + 1)                         # ghost layer at index 0 and n + 1
                inside.append((c >= 0) & (c < n[d]))
        def at(arr_d):       # arr_d: [k][j][i] ghosted
            sel = [slice(1, 2)] * 3 if dim < 3 else None
            ix = [np.array([0])] * 3
            for d in range(dim):
                ix[2 - d] = idx[d]
            if dim < 3:
                ix[0] = np.array([0])
            return arr_d[np.ix_(*ix)]
        def mask(skip):
            m = np.ones(shape, dtype=float)
            for e in range(dim):
                if e == skip:
                    continue
                v = inside[e].astype(float)
                sh = [1, 1, 1]
                sh[2 - e] = nn[e]
                m = m * v.reshape(sh)
            return m
        for d in range(dim):
            sgn = 1.0 if a[d] else -1.0
            rhs += (fac / dx[d]) * sgn * at(Vg[d]) * mask(d)
        if S_cc is not None:
            Sg = np.zeros(Vg.shape[1:])
            inner = tuple(slice(1, -1) if (2 - ax) < dim else slice(None) for ax in range(3))
            Sg[inner] = S_cc
            # periodic directions: the source wraps (FillBoundary inside compRHS); beyond physical faces it is 0
            for d in range(dim):
                if per[d]:
                    ax = 2 - d
                    lo_g = [slice(None)] * 3
                    hi_g = [slice(None)] * 3
                    lo_g[ax], hi_g[ax] = 0, -1
                    src_lo, src_hi = [slice(None)] * 3, [slice(None)] * 3
                    src_lo[ax], src_hi[ax] = -2, 1
                    Sg[tuple(lo_g)] = Sg[tuple(src_lo)]
                    Sg[tuple(hi_g)] = Sg[tuple(src_hi)]
            rhs += (0.5 ** dim) * at(Sg) * mask(-1)
    for d in range(dim):       # mlndlap_impose_neumann_bc; Dirichlet nodes carry no equation
        sh = [1, 1, 1]
        sh[2 - d] = nn[d]
        f = np.ones(nn[d])
        if not per[d]:
            if bc[d][0] == NEUMANN:
                f[0] = 2.0
            if bc[d][1] == NEUMANN:
                f[-1] = 2.0
            if bc[d][0] == DIRICHLET:
                f[0] = 0.0
            if bc[d][1] == DIRICHLET:
                f[-1] = 0.0
        rhs = rhs * f.reshape(sh)
    ops = [_ops_1d(n[d], dx[d], bc[d][0], bc[d][1]) for d in range(dim)]
    ident = [sp.identity(1, format="csr")] * 3

    def kron3(mats):     # index ((k) nn1 + j) nn0 + i  ->  kron(z, kron(y, x))
        full = [sp.identity(1, format="csr")] * 3
        for d in range(dim):
            full[d] = mats[d]
        return sp.kron(full[2], sp.kron(full[1], full[0], format="csr"), format="csr")
    L = None
    for d in range(dim):
        term = kron3([ops[e][0] if e == d else ops[e][1] for e in range(dim)])
        L = -term if L is None else L - term
    wfull = np.ones(shape)
    free_mask = np.ones(shape, dtype=bool)
    for d in range(dim):
        sh = [1, 1, 1]
        sh[2 - d] = nn[d]
        wfull = wfull * ops[d][3].reshape(sh)
        fm = np.zeros(nn[d], dtype=bool)
        fm[ops[d][2]] = True
        free_mask = free_mask & fm.reshape(sh)
    free = np.flatnonzero(free_mask.ravel())
    w = wfull.ravel()[free]
    A = sp.diags(w) @ L[free][:, free]                     # symmetric: the assembled element matrices
    b = w * rhs.ravel()[free]
    singular = all(bc[d][0] != DIRICHLET and bc[d][1] != DIRICHLET for d in range(dim))
    if singular:
        col = sp.csr_matrix(w.reshape(-1, 1))
        aug = sp.bmat([[A, col], [col.T, None]], format="csc")
        x = spl.spsolve(aug, np.concatenate([b, [0.0]]))[:-1]
    else:
        x = spl.spsolve(A.tocsc(), b)
    phi = np.zeros(int(np.prod(shape)))
    phi[free] = x
    phi = phi.reshape(shape)
    # V -= G phi at the cell centres, phi_cc = average of the cell's nodes (getFluxes, average_node_to_cellcenter)
    cell = [np.arange(n[d]) for d in range(dim)]
    Vn = np.array([Vg[d][tuple(slice(1, -1) if (2 - ax) < dim else slice(None) for ax in range(3))] for d in range(dim)])
    grad = [np.zeros(Vn.shape[1:]) for _ in range(dim)]
    avg = np.zeros(Vn.shape[1:])
    for corner in range(1 << dim):
        a = [(corner >> d) & 1 for d in range(dim)]
        ix = [np.array([0])] * 3
        for d in range(dim):
            c = cell[d] + a[d]
            ix[2 - d] = c % nn[d] if per[d] else c
        p = phi[np.ix_(*ix)]
        avg += p
        for d in range(dim):
            grad[d] += (1.0 if a[d] else -1.0) * p
    for d in range(dim):
        Vn[d] = Vn[d] - (fac / dx[d]) * grad[d]
    return Vn, avg * (0.5 ** dim)


def project_divergence_field_bc(Fg, ng, rhos, rs, dx, dim, cd_scale, periodic, bc_B, bc_D):
    """Elliptic::project_divergence_field (MFP_elliptic.cpp:667-782) on a domain with physical faces.  Fg: the field
    components [10] with `ng` filled ghost layers in every used dimension (FillPatch, :897-904); rhos: the species'
    densities on the valid cells; bc_B / bc_D: physical face types [2 * dim + side] (FieldState::bc_names values).
    Returns the new field on the valid cells."""
    phys = {1: DIRICHLET, 2: NEUMANN, 3: NEUMANN}      # set_field_bcs (:93-119); 4 ('asymmetry') -> reflect_odd: refused
    def faces(t):
        out = []
        for d in range(dim):
            if periodic[d]:
                out.append((PERIODIC, PERIODIC))
            else:
                if t[2 * d] not in phys or t[2 * d + 1] not in phys:
                    raise NotImplementedError("divergence projection: faces must be inflow / outflow / symmetry")
                out.append((phys[t[2 * d]], phys[t[2 * d + 1]]))
        return out
    one = tuple(slice(ng - 1, -(ng - 1) if ng > 1 else None) if (2 - ax) < dim else slice(None) for ax in range(3))
    valid = tuple(slice(ng, -ng) if (2 - ax) < dim else slice(None) for ax in range(3))
    F = np.array([Fg[c][valid] for c in range(10)])
    Vb = np.array([Fg[3 + d][one] for d in range(dim)])
    Vn, F[7] = solve_divergence_bc(Vb, dx, dim, faces(bc_B))
    F[3:3 + dim] = Vn
    cd = np.zeros_like(F[0])
    for rho, r in zip(rhos, rs):
        cd = cd + rho * r
    S = (cd / F[9]) * (-cd_scale)
    Vd = np.array([Fg[0 + d][one] for d in range(dim)])
    Vn, F[6] = solve_divergence_bc(Vd, dx, dim, faces(bc_D), S)
    F[0:dim] = Vn
    return F


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
        self._dynamic = [s for s in range(len(problem.state_names)) if problem.dynamic_entries(s)]
        self._dynamic_t = [s for s in self._dynamic if problem.dynamic_depends_on_time(s)]
        for s in self._dynamic:
            self._set_dynamic(s, 0.0)

    def _set_dynamic(self, s, t):
        for comp, values in self.p.dynamic_tables(s, self.TNG, t):
            lib().orc_set_dynamic(self.h, C.c_int(s), C.c_int(comp), C.c_int(self.TNG), _p(values), C.c_long(values.size))

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
        for s in self._dynamic_t:
            self._set_dynamic(s, self.time + dt if self.p.scheme == "strang" else self.time)
        self.time += dt
        lib().orc_clear_error()
        lib().orc_advance(self.h, C.c_double(dt))
        if lib().orc_error_flag():
            raise FloatingPointError("Primitive values outside of physical bounds!!")
        for fs, species in self.p.elliptic_actions():      # act->apply_change (MFP_advance.cpp:49)
            if self.p.dim < 2:
                continue                                   # MLNodeLaplacian not implemented for 1D (MFP_elliptic.cpp:640-650)
            r = self.p.ref
            gas = lambda s: self.p.cfg["states"][self.p.state_names[s]]["gas"]

            def r_of(s):     # get_charge_from_cons / get_mass_from_cons (HydroState::calc_current_and_charge, MFP_hydro.cpp:1720-1721)
                from cerberus_amd import config
                U = self.get_state(s)
                al = [U[5 + i] / U[0] for i in range(config.gas_n_tracers(gas(s)))]
                return config.gas_charge_from_alpha(gas(s), al) / config.gas_mass_from_alpha(gas(s), al)
            cd_scale = r["Larmor"] / (r["Debye"] ** 2 * r["lightspeed"])
            if all(self.p.periodic[:self.p.dim]):
                F = project_divergence_field(self.get_state(fs), [self.get_state(s)[0] for s in species],
                                             [r_of(s) for s in species], self.p.dx, self.p.dim, cd_scale)
            else:
                bc_B, bc_D = self.p.elliptic_face_types(fs)
                F = project_divergence_field_bc(self.fill_patch(fs), abi_ngrow(self.p, fs), [self.get_state(s)[0] for s in species],
                                                [r_of(s) for s in species], self.p.dx, self.p.dim, cd_scale,
                                                self.p.periodic, bc_B, bc_D)
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
