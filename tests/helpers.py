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


def rel_linf(a, b):
    """relative L-infinity per state: max|a-b| / max|b| (north_star: <= 1e-12 on conserved variables)."""
    den = np.abs(b).max()
    return np.abs(a - b).max() / (den if den > 0 else 1.0)
