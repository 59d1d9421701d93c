"""Physical boundary name -> per-component mathematical ghost rule.

Mirror of the tables and set_*_bc functions in the reference's
Source/modules/boundary/MFP_bc.cpp:11-31 (hydro) and :99-109 (field), and of
the way HydroState / FieldState apply them per component
(hydro/MFP_hydro.cpp:35-49,444-463; field/MFP_field.cpp:7-18,36-40,296-359,
405-424).  Output: bc[comp][dim][side] of cmfp_state_desc.
"""
from . import abi

INT, FO, EVEN, ODD = abi.BC_INT_DIR, abi.BC_FOEXTRAP, abi.BC_REFLECT_EVEN, abi.BC_REFLECT_ODD

# hydro physical types (MFP_hydro.cpp:35-40): index into the tables below
HYDRO_BC_NAMES = {"interior": 0, "inflow": 1, "outflow": 2, "symmetry": 3, "slipwall": 4, "noslipwall": 5}
SCALAR_BC = [INT, FO, FO, EVEN, EVEN, EVEN, ODD]        # MFP_bc.cpp:11-17
NORM_VEL_BC = [INT, FO, FO, ODD, ODD, ODD]              # :19-24
TANG_VEL_BC = [INT, FO, FO, EVEN, EVEN, ODD]            # :26-31

# field physical types (MFP_field.cpp:36-40)
FIELD_BC_NAMES = {"interior": 0, "inflow": 1, "outflow": 2, "symmetry": 3, "asymmetry": 4}
NORM_D_BC = [INT, FO, FO, ODD, EVEN]                    # MFP_bc.cpp:99-100
TANG_D_BC = [INT, FO, FO, EVEN, ODD]                    # :102-103
NORM_B_BC = [INT, FO, FO, EVEN, ODD]                    # :105-106
TANG_B_BC = [INT, FO, FO, ODD, EVEN]                    # :108-109

DIR_NAME = ["x", "y", "z"]
SIDE_NAME = ["lo", "hi"]


def _get(bc_def, d, side, key, default):
    try:
        return bc_def[DIR_NAME[d]][SIDE_NAME[side]].get(key, default)
    except (KeyError, TypeError, AttributeError):
        return default


def hydro_bc(bc_def, dim, periodic, n_tracers=0):
    """bc[comp 0..4 + n_tracers][d][side] for {rho, x_mom, y_mom, z_mom, nrg, tracer_i}."""
    out = [[[abi.BC_BOGUS, abi.BC_BOGUS] for _ in range(abi.MAX_DIM)] for _ in range(abi.MAX_COMP)]
    for d in range(dim):
        for side in range(2):
            name = _get(bc_def, d, side, "fill_hydro_bc", "outflow")  # MFP_hydro.cpp:342-345
            if name not in HYDRO_BC_NAMES:
                raise ValueError(f"unknown fill_hydro_bc '{name}'")
            t = 0 if periodic[d] else HYDRO_BC_NAMES[name]      # MFP_hydro.cpp:454-459
            out[0][d][side] = SCALAR_BC[t]
            out[4][d][side] = SCALAR_BC[t]
            for i in range(n_tracers):                          # bc_set.push_back(&set_scalar_bc), MFP_hydro.cpp:323
                out[5 + i][d][side] = SCALAR_BC[t]
            for v in range(3):                                  # set_{x,y,z}_vel_bc MFP_bc.cpp:43-89
                out[1 + v][d][side] = NORM_VEL_BC[t] if v == d else TANG_VEL_BC[t]
    return out


def field_bc(bc_def, dim, periodic):
    """bc[comp 0..9][d][side] for {Dx,Dy,Dz,Bx,By,Bz,phi,psi,mu,ep}."""
    out = [[[abi.BC_BOGUS, abi.BC_BOGUS] for _ in range(abi.MAX_DIM)] for _ in range(abi.MAX_COMP)]

    def phys(d, side, key):
        name = _get(bc_def, d, side, key, "outflow")             # MFP_field.cpp:323-326
        if name not in FIELD_BC_NAMES:
            raise ValueError(f"unknown {key} '{name}'")
        return 0 if periodic[d] else FIELD_BC_NAMES[name]        # MFP_field.cpp:415-420

    for d in range(dim):
        for side in range(2):
            tD, tB = phys(d, side, "fill_D_bc"), phys(d, side, "fill_B_bc")
            for v in range(3):                                   # set_{x,y,z}_{D,B}_bc MFP_bc.cpp:111-205
                out[0 + v][d][side] = NORM_D_BC[tD] if v == d else TANG_D_BC[tD]
                out[3 + v][d][side] = NORM_B_BC[tB] if v == d else TANG_B_BC[tB]
            out[6][d][side] = SCALAR_BC[phys(d, side, "fill_phi_bc")]
            out[7][d][side] = SCALAR_BC[phys(d, side, "fill_psi_bc")]
            out[8][d][side] = SCALAR_BC[phys(d, side, "fill_mu_bc")]
            out[9][d][side] = SCALAR_BC[phys(d, side, "fill_ep_bc")]
    return out


HYDRO_PRIM_NAMES = ["rho", "x_vel", "y_vel", "z_vel", "p", "T", "gamma", "cp"]   # MFP_hydro.cpp:21-30
FIELD_NAMES = ["x_D", "y_D", "z_D", "x_B", "y_B", "z_B", "phi", "psi", "mu", "ep"]


def face_values(sd, dim, periodic, n_tracers=0):
    """Per-face Dirichlet entries of a state definition: {(d, side): {carried primitive index: value}}
    for every primitive name given a number or a function in states.<name>.bc.<dim>.<side>
    (HydroState MFP_hydro.cpp:347-366, 383-409; FieldState MFP_field.cpp:340-352).  A face with at
    least one entry is the reference's where_is_inflow = ext_dir (MFP_fillbc.cpp:72-87).
    fill_hydro_bc = 'inflow' needs every primitive defined (:361-365)."""
    out = {}
    bc_def = sd.get("bc", {})
    hydro = sd["type"] == "hydro"
    for d in range(dim):
        for side in range(2):
            try:
                face = bc_def[DIR_NAME[d]][SIDE_NAME[side]]
            except (KeyError, TypeError):
                continue
            vals = {}
            if hydro:
                for n, name in enumerate(HYDRO_PRIM_NAMES):
                    if name in face and n < 5:            # T, gamma, cp never reach the conserved result
                        vals[n] = face[name]
                    if face.get("fill_hydro_bc") == "inflow" and name not in face:
                        raise ValueError("Setting 'fill_hydro_bc = inflow' requires all primitive variables to be "
                                         f"defined, '{name}' is not defined")
                alpha = face.get("alpha")
                if alpha is not None:
                    alpha = list(alpha) if isinstance(alpha, (list, tuple)) else [alpha] * n_tracers
                    if len(alpha) < n_tracers:
                        raise ValueError("Not enough boundary conditions specified for alpha")
                    for i in range(n_tracers):
                        vals[5 + i] = alpha[i]
                elif face.get("fill_hydro_bc") == "inflow" and n_tracers:
                    raise ValueError("Setting 'fill_hydro_bc = inflow' requires all primitive variables to be "
                                     "defined, 'alpha' is not defined")
            else:
                for n, name in enumerate(FIELD_NAMES):
                    if name in face:
                        vals[n] = face[name]
            if vals:
                if periodic[d]:
                    raise ValueError("boundary values given on a periodic face")
                out[(d, side)] = vals
    return out
