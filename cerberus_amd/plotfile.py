"""AMReX plotfile output of the HBM-resident state.

Restates MFP::writePlotFile / getPlotData / writePlotFilePost (MFP_io.cpp:118-303, 305-422,
467-545) for the single-level uniform grid of this build, so that the reference's own
post-processing (vis/get_boxlib.py `ReadBoxLib`, the `check.py` scripts of Exec/testing) reads
what the GPU path produced:

    <dir>/Header                 HyperCLaw-V1.1 text header (MFP_io.cpp:320-355, 382-400)
    <dir>/Level_0/Cell_H         VisMF header, version 1: box array, FabOnDisk, per-box min / max
    <dir>/Level_0/Cell_D_0000r   one file per rank: "FAB ..." text line + fp64 data per box,
                                 i fastest, component slowest (the FArrayBox layout)
    <dir>/info                   JSON: reference scales (MFP::write_info :526-545) and per-state
                                 metadata (State / EulerianState / HydroState / FieldState /
                                 ThermallyPerfectGas ::write_info)

The VisMF layout comes from the reference's reader (vis/get_boxlib.py:48-119, 683-754), which is
what must be able to parse it; the AMReX writer itself is not in the reference tree.
Variables follow plot.variables = {'all'} (MFP_config.cpp:326-335): `cost` and, for every state,
its get_plot_output_names() as `<var>-<state>`, ordered as std::map orders its keys.  `cost` (the
reference's per-box wall time) is written as zero.  This is host-side I/O at the end of a step:
the state is downloaded once through cmfp_state_download.
"""
import json
import os

import numpy as np

from . import config

HYDRO_CONS = ["rho", "x_mom", "y_mom", "z_mom", "nrg"]          # MFP_hydro.cpp:13-19
HYDRO_PRIM = config.HYDRO_PRIM                                     # MFP_hydro.cpp:21-30
FIELD_CONS = config.FIELD_CONS                                     # MFP_field.cpp:20-31
STATE_TYPE_IDX = {"hydro": 0, "field": 1}                          # State::StateType, MFP_state.H:31
FAB_DESCRIPTOR = "((8, 64 11 52 0 1 12 0 1023),(8, 1 2 3 4 5 6 7 8))"  # IEEE little-endian double


def state_plot_names(P, s):
    """get_plot_output_names(): HydroState MFP_hydro.cpp:569-589, FieldState MFP_field.cpp:68-79."""
    sd = P.cfg["states"][P.state_names[s]]
    if sd["type"] == "hydro":
        names = HYDRO_CONS + HYDRO_PRIM + ["charge", "mass", "gamma"]
        for i in range(config.gas_n_tracers(sd["gas"])):     # get_multicomp_name, MFP_hydro.cpp:583-586
            names += [f"tracer_{i}", f"alpha_{i}"]
        return names
    return list(FIELD_CONS)


def plot_variable_names(P):
    """plot.variables = {'all'}: keys of MFP::plot_variables in std::map order (MFP_config.cpp:326-343)."""
    names = {"cost"}
    for s, sname in enumerate(P.state_names):
        for v in state_plot_names(P, s):
            names.add(f"{v}-{sname}")
    return sorted(names)     # byte-wise, like std::map<std::string, ...>


def state_plot_data(P, s, U):
    """HydroState::get_plot_output (MFP_hydro.cpp:591-724) / FieldState::get_plot_output: every
    plot variable of state s from its conserved data U[comp, k, j, i]."""
    sname = P.state_names[s]
    sd = P.cfg["states"][sname]
    out = {}
    if sd["type"] != "hydro":
        for n, v in enumerate(FIELD_CONS):
            out[f"{v}-{sname}"] = U[n]
        return out
    gas = sd["gas"]
    nt = config.gas_n_tracers(gas)
    ez = float(sd.get("effective_zero", 1e-14))
    ok = U[0] >= ez                                   # :700 cells below effective_zero stay 0
    rho = np.where(ok, U[0], 1.0)
    # gas constants of the cell: get_*_from_cons with alpha_i = tracer_i / rho (numbers where the species agree)
    ac = [U[5 + i] / rho for i in range(nt)]
    mass, charge = config.gas_mass_from_alpha(gas, ac), config.gas_charge_from_alpha(gas, ac)
    gamma, cp = config.gas_gamma_from_alpha(gas, ac), config.gas_cp_from_alpha(gas, ac)
    # ThermallyPerfectGas::cons2prim, MFP_thermally_perfect_gas.cpp:180-218
    rhoinv = 1.0 / rho
    u, v, w = U[1] * rhoinv, U[2] * rhoinv, U[3] * rhoinv
    ke = 0.5 * rho * (u * u + v * v + w * w)
    p = (U[4] - ke) * (gamma - 1)
    T = p * rhoinv * mass
    one = np.ones_like(rho)
    vals = dict(zip(HYDRO_CONS, U[:5]))
    vals.update({"x_vel": u, "y_vel": v, "z_vel": w, "p": p, "T": T, "gamma": gamma * one, "cp": cp * one,
                 "charge": charge * one, "mass": mass * one})
    for i in range(nt):                               # :622-645, alpha = tracer * rhoinv (:212-214)
        vals[f"tracer_{i}"] = U[5 + i]
        vals[f"alpha_{i}"] = U[5 + i] * rhoinv
    for k, a in vals.items():
        out[f"{k}-{sname}"] = np.where(ok, a, 0.0)
    return out


def state_info(P, s):
    """State::write_info (MFP_state.cpp:203-212) + EulerianState (:1730-1741) + HydroState
    (MFP_hydro.cpp:1738-1760) / FieldState (MFP_field.cpp:553-563) + gas (MFP_hydro_gas.cpp:232-239,
    MFP_thermally_perfect_gas.cpp:334-342)."""
    sname = P.state_names[s]
    sd = P.cfg["states"][sname]
    recon = sd["reconstruction"]
    ngrow = {"constant": 1, "O6": 4, "WENO": 4}.get(recon, 2)
    js = {"name": sname, "global_idx": s, "data_idx": s, "num_grow": ngrow, "type": sd["type"],
          "type_idx": STATE_TYPE_IDX[sd["type"]]}
    if sd["type"] == "hydro":
        gas = sd["gas"]
        js["cons_names"] = list(HYDRO_CONS)
        nsp = config.gas_n_tracers(gas) + 1
        js["gas"] = {k: config.gas_list(gas, k) for k in ("mass", "charge", "gamma")}
        assert all(len(v) == nsp for v in js["gas"].values())
        js["gas"].update({"comp_names": [], "type": gas.get("type", "thermally_perfect")})
        if "viscosity" in sd:
            js["viscosity"] = dict(sd["viscosity"])
    else:
        js["cons_names"] = list(FIELD_CONS)
    return js


def global_info(P):
    """MFP::write_info, MFP_io.cpp:526-545."""
    r = P.ref
    g = dict(config.DEFAULTS)
    g.update({k: P.cfg[k] for k in config.DEFAULTS if k in P.cfg})
    m_ref = g["ref_mass"]
    return {"num_state": len(P.state_names), "x_ref": r["x_ref"], "n_ref": r["rho_ref"] / m_ref, "m_ref": m_ref,
            "rho_ref": r["rho_ref"], "T_ref": r["T_ref"], "u_ref": r["u_ref"], "n0": r["n0"],
            "lightspeed": r["lightspeed"], "beta": r["beta"], "skin_depth": r["skin_depth"],
            "Larmor": r["Larmor"], "Debye": r["Debye"]}


def _box_str(lo, hi, dim):
    z = ",".join("0" for _ in range(dim))
    return "((%s) (%s) (%s))" % (",".join(str(int(v)) for v in lo[:dim]), ",".join(str(int(v)) for v in hi[:dim]), z)


def chop(lo, n, dim, max_grid_size):
    """Boxes of at most max_grid_size cells per side covering the block [lo, lo + n) (the reference's
    amr.max_grid_size chopping); None: the block as one box."""
    if not max_grid_size:
        return [(list(lo), [lo[d] + n[d] - 1 for d in range(3)])]
    cuts = []
    for d in range(3):
        if d >= dim:
            cuts.append([(lo[d], lo[d] + n[d] - 1)])
            continue
        nb = -(-n[d] // max_grid_size)
        edges = [lo[d] + (n[d] * b) // nb for b in range(nb + 1)]
        cuts.append([(edges[b], edges[b + 1] - 1) for b in range(nb)])
    return [([x[0], y[0], z[0]], [x[1], y[1], z[1]]) for z in cuts[2] for y in cuts[1] for x in cuts[0]]


def write_fabs(path, fname, P, names, data, block_lo, block_n, max_grid_size=None):
    """Writes this rank's boxes into <path>/Level_0/<fname>.  data: {plot name: array[k, j, i]} over
    the rank's block.  Returns [(lo, hi, fname, offset, mins, maxs)] for the VisMF header."""
    dim = P.dim
    os.makedirs(os.path.join(path, "Level_0"), exist_ok=True)
    recs = []
    with open(os.path.join(path, "Level_0", fname), "wb") as f:
        for lo, hi in chop(block_lo, block_n, dim, max_grid_size):
            sl = tuple(slice(lo[d] - block_lo[d], hi[d] - block_lo[d] + 1) for d in (2, 1, 0))
            off = f.tell()
            f.write(("FAB %s%s %d\n" % (FAB_DESCRIPTOR, _box_str(lo, hi, dim), len(names))).encode())
            mins, maxs = [], []
            for nm in names:
                a = np.ascontiguousarray(data[nm][sl], dtype="<f8")    # C order [k][j][i] = i fastest
                f.write(a.tobytes())
                mins.append(float(a.min()))
                maxs.append(float(a.max()))
            recs.append((lo, hi, fname, off, mins, maxs))
    return recs


def write_headers(path, P, names, recs, time, step):
    """Header (MFP_io.cpp:320-355, 382-400), Level_0/Cell_H and info (:467-503) from the box records
    of all ranks."""
    dim = P.dim
    rep = lambda v: repr(float(v))
    with open(os.path.join(path, "Header"), "w") as f:
        f.write("HyperCLaw-V1.1\n%d\n" % len(names))
        for nm in names:
            f.write(nm + "\n")
        f.write("%d\n%s\n0\n" % (dim, rep(time)))
        f.write(" ".join(rep(P.prob_lo[d]) for d in range(dim)) + " \n")
        f.write(" ".join(rep(P.prob_hi[d]) for d in range(dim)) + " \n")
        f.write("\n")                                                   # refinement ratios: none at max_level = 0
        f.write(_box_str([0, 0, 0], [P.n_cell[d] - 1 for d in range(3)], dim) + " \n")
        f.write("%d \n" % step)
        f.write(" ".join(rep(P.dx[d]) for d in range(dim)) + " \n")
        f.write("0\n0\n")                                               # coord_sys, bndry data
        f.write("0 %d %s\n%d\n" % (len(recs), rep(time), step))
        for lo, hi, *_ in recs:                                         # RealBox of every grid
            for d in range(dim):
                f.write("%s %s\n" % (rep(P.prob_lo[d] + lo[d] * P.dx[d]), rep(P.prob_lo[d] + (hi[d] + 1) * P.dx[d])))
        f.write("Level_0/Cell\n")
    with open(os.path.join(path, "Level_0", "Cell_H"), "w") as f:
        f.write("1\n1\n%d\n0\n" % len(names))                           # version, how, ncomp, nghost
        f.write("(%d 0\n" % len(recs))
        for lo, hi, *_ in recs:
            f.write(_box_str(lo, hi, dim) + "\n")
        f.write(")\n%d\n" % len(recs))
        for _, _, fname, off, _, _ in recs:
            f.write("FabOnDisk: %s %d\n" % (fname, off))
        f.write("\n")
        for idx in (4, 5):                                              # per-box minima, then maxima
            f.write("%d,%d\n" % (len(recs), len(names)))
            for r in recs:
                f.write(",".join("%.17e" % v for v in r[idx]) + ",\n")
            f.write("\n")
    info = global_info(P)
    for s in range(len(P.state_names)):
        info["state_%d" % s] = state_info(P, s)
    info["version"] = "cerberus_amd"
    info["AMReX_version"] = "none (plotfile written by cerberus_amd/plotfile.py)"
    with open(os.path.join(path, "info"), "w") as f:
        json.dump(info, f, indent=4)
        f.write("\n")


def write_plotfile(path, P, states, time, step, block_lo=None, block_n=None, max_grid_size=None, rank=0, world=1,
                   gather=None):
    """states[s]: conserved data [comp, k, j, i] of this rank's block.  With world > 1 every rank
    writes its own Cell_D file and `gather(obj) -> list` (e.g. torch.distributed.all_gather_object
    wrapped) collects the box records; rank 0 writes the headers."""
    block_lo = [0, 0, 0] if block_lo is None else list(block_lo)
    block_n = list(P.n_cell) if block_n is None else list(block_n)
    names = plot_variable_names(P)
    data = {"cost": np.zeros(tuple(block_n[::-1]))}
    for s in range(len(P.state_names)):
        data.update(state_plot_data(P, s, np.asarray(states[s])))
    recs = write_fabs(path, "Cell_D_%05d" % rank, P, names, data, block_lo, block_n, max_grid_size)
    if world > 1:
        allrecs = [r for part in gather(recs) for r in part]
    else:
        allrecs = recs
    if rank == 0:
        write_headers(path, P, names, allrecs, time, step)
    return names
