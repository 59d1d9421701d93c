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


def rel_linf(a, b):
    """relative L-infinity per state: max|a-b| / max|b| (north_star: <= 1e-12 on conserved variables)."""
    den = np.abs(b).max()
    return np.abs(a - b).max() / (den if den > 0 else 1.0)
