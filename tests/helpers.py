"""Shared helpers of the parity tests."""
import numpy as np

from cerberus_amd import abi, config, problems

# reduced-size versions of the BASELINE.json configurations: (name, cfg, steps)
def small_cases():
    return [
        ("brio_wu_1d", problems.brio_wu_two_fluid_1d(256), 20),
        ("kh_2d", problems.kelvin_helmholtz_2d(48), 10),
        ("ot2f_2d", problems.orszag_tang_two_fluid_2d(48), 8),
        ("gem_2d", problems.gem_reconnection_2d(48, 24), 6),
        ("rmi_3d", problems.richtmyer_meshkov_3d(24, 12, 12), 4),
    ]


# the sizes SURVEY.md section 8(d) names for the parity runs: config 1 at 2048 cells x 100 steps, configs 2 / 3 at
# 256^2 x 20, the GEM block 256 x 128 x 20, RMI 64^3 x 10 (wider than one 60-column marching strip, several
# 14 x 14 tiles per XCD band, several z-segments)
def survey_cases():
    return [
        ("brio_wu_1d@2048", problems.brio_wu_two_fluid_1d(2048), 100),
        ("kh_2d@256", problems.kelvin_helmholtz_2d(256), 20),
        ("ot2f_2d@256", problems.orszag_tang_two_fluid_2d(256), 20),
        ("gem_2d@256x128", problems.gem_reconnection_2d(256, 128), 20),
        ("rmi_3d@64", problems.richtmyer_meshkov_3d(64), 10),
    ]


# Per-component relative L-infinity needs one rule for components that are small against the values they are computed
# from.  In fp64 (rounding unit u = 1.1e-16) a component that is updated from quantities of magnitude T picks up an
# absolute difference of a few u T between ANY two evaluation orders (FAST against STRICT, contracted against not):
# measured on the HIP path after 4-20 steps (tools/comp_scales.py, profiles/r03/comp_scales.json) every component of
# every state differs by 1e-17 ... 7e-16 of its state's largest value, whatever its own size - D_x of the Orszag-Tang
# field (a difference of the two species' currents, 1e-8 of |B|) by 1e-17, psi (pure cancellation noise, 4e-14) by
# 5e-16.  A relative bound of 1e-12 on a component below 1e-3 of its state's scale would ask for less than that
# rounding unit, so such components are measured against 1e-3 of the state's largest value: the bound then reads
# |a - b| <= 1e-12 * max(max|b_c|, 1e-3 max|b|), i.e. an absolute floor of 1e-15 max|b| (9 ulp of the state's scale).
ROUNDING_FLOOR = 1e-3


def rel_linf_per_comp(a, b, floor=ROUNDING_FLOOR):
    """relative L-infinity of every conserved component of a state [comp, ...]: max|a_c - b_c| / max|b_c|
    (north_star: <= 1e-12 on conserved variables), components smaller than `floor` times the state's largest
    value measured against that (ROUNDING_FLOOR above)."""
    a, b = np.asarray(a), np.asarray(b)
    nc = b.shape[0]
    top = np.abs(b).max()
    top = top if top > 0 else 1.0
    out = np.empty(nc)
    for c in range(nc):
        den = max(np.abs(b[c]).max(), floor * top)
        out[c] = np.abs(a[c] - b[c]).max() / den
    return out


def rel_linf(a, b):
    """the largest per-component relative L-infinity error of a state (rel_linf_per_comp)."""
    return float(rel_linf_per_comp(a, b).max())
