"""The exact path bench.py times, at the size it times it.

bench.py's headline is BASELINE configs[2]: two-fluid Orszag-Tang at 4096^2 in FAST arithmetic with default settings.
Above 2^23 cells that step is: x ghost-fill launch, y wrapped inside the marching kernels, Strang source stage A as
the CTU kernels' prologue and stage C as the field kernel's epilogue (mfp_context.cpp, `wrap_flag`).  No other test
ran that combination (wrap-both or wrap-none, prologue up to 2048 x 520).  Here it is checked where it is measured:

* the step reports that path (cmfp_last_source_fusion == 3, cmfp_last_wrapped_dims == 2, fused kernels);
* its result after N steps is within the north-star bound 1e-12 - per conserved component, max|a - b| / max|b| -
  of the same problem in STRICT arithmetic on the stage-by-stage kernels (the kernels that mirror the reference's
  FAB-level functions one to one and are bit-identical to the oracle at every size the oracle reaches);
* species mass is conserved to rounding on the periodic domain (through the walls of the GEM block too);
* the same for the GEM block 4096 x 2048 (BASELINE configs[3] per GPU): y ghost-fill launch, x wrapped.
* the other two single-GPU legs of the bench at their sizes: Kelvin-Helmholtz 2048^2 (configs[1]) and Richtmyer-Meshkov
  256^3 (configs[4] reduced; 512^3 runs the same kernels) — FAST fused against STRICT staged, per component.
The oracle itself is too slow at this size (0.6 M cell-updates/s per core); a small oracle case of the y-only wrap
is in test_gpu_wrap.py."""
import numpy as np
import pytest

from cerberus_amd import abi, config, problems
from helpers import rel_linf_per_comp

pytestmark = pytest.mark.gpu

TOL = 1e-12        # north_star: relative L-inf per conserved component after N steps
STEPS = 4


def _run(cfg, mode, fused, steps):
    from cerberus_amd.mfp import MFP
    P = config.Problem(cfg, fp_mode=mode)
    m = MFP(P, device=0, fused=fused)
    m.init_data()
    dts = m.run(steps)
    info = {"path": m.last_advance_path(), "fusion": m.last_source_fusion(), "wrapped": m.last_wrapped_dims()}
    return P, m, dts, info


def _check(make, wrapped_expected, walls):
    P, m, dts, info = _run(make(), abi.FP_FAST, True, STEPS)
    assert info["path"] == 1, "the fused marching kernels"
    assert info["fusion"] == 3, "stage A and stage C inside the CTU kernels"
    assert info["wrapped"] == wrapped_expected, info
    fast = [m.download(s) for s in range(m.n_states)]
    m.close()
    # species mass: the flux form telescopes over the periodic directions and the wall fluxes carry no mass
    for s, name in enumerate(P.state_names):
        if name == "field":
            assert np.all(fast[s][8] == 1.0) and np.all(fast[s][9] == 1.0)
            continue
        m0 = P.initial_data(s)[0].sum(dtype=np.longdouble)
        m1 = fast[s][0].sum(dtype=np.longdouble)
        assert abs(m1 - m0) <= (1e-11 if walls else 1e-12) * abs(m0), (name, float((m1 - m0) / m0))
    P2, r, dts_ref, info_ref = _run(make(), abi.FP_STRICT, False, STEPS)
    assert info_ref["path"] == 0 and info_ref["fusion"] == 0 and info_ref["wrapped"] == 0
    np.testing.assert_allclose(dts, dts_ref, rtol=1e-12)
    worst = 0.0
    for s in range(r.n_states):
        ref = r.download(s)
        e = rel_linf_per_comp(fast[s], ref)
        assert e.max() <= TOL, (P.state_names[s], e.tolist())
        worst = max(worst, float(e.max()))
        fast[s] = None
    r.close()
    return worst


def _populated_ot(n):
    """Orszag-Tang with every component of every state at >= 1e-2 of its state's scale, so that each is held to
    1e-12 of its OWN largest value (no component falls under the rounding floor of helpers.rel_linf_per_comp)"""
    cfg = problems.orszag_tang_two_fluid_2d(n)
    for name in ("ion", "electron"):
        cfg["states"][name]["value"]["z_vel"] = lambda d: 0.3 * np.sin(d["x"] + d["y"])
    v = cfg["states"]["field"]["value"]
    v["x_D"] = lambda d: 0.02 * np.sin(d["y"])
    v["y_D"] = lambda d: 0.02 * np.cos(d["x"])
    v["z_B"] = lambda d: 0.2 * np.cos(d["x"] + d["y"])
    v["phi"] = lambda d: 0.02 * np.sin(d["x"])
    v["psi"] = lambda d: 0.02 * np.cos(d["y"])
    return cfg


def test_orszag_tang_4096_every_component_populated():
    """the same path with initial data in which no component is small: every conserved component within 1e-12 of
    its own largest value (checked: none of them is measured against the floor)"""
    make = lambda: _populated_ot(4096)
    P = config.Problem(make(), fp_mode=abi.FP_STRICT)
    for s in range(len(P.state_names)):
        u = P.initial_data(s)
        top = np.abs(u).max()
        small = [c for c in range(u.shape[0]) if np.abs(u[c]).max() < 1e-2 * top]
        assert not small, (P.state_names[s], small)
    worst = _check(make, 2, walls=False)
    print(f"ot2f 4096^2 populated: worst per-component rel L-inf {worst:.2e}")


def test_orszag_tang_4096_fast_default_path():
    worst = _check(lambda: problems.orszag_tang_two_fluid_2d(4096), 2, walls=False)      # bit 1: y wrapped, x filled
    print(f"ot2f 4096^2 FAST (bench path) vs STRICT staged after {STEPS} steps: worst per-component rel L-inf {worst:.2e}")


def test_gem_block_4096x2048_fast_default_path():
    worst = _check(lambda: problems.gem_reconnection_2d(4096, 2048), 1, walls=True)      # bit 0: x wrapped, walls in y
    print(f"gem 4096x2048 FAST (bench path) vs STRICT staged after {STEPS} steps: worst per-component rel L-inf {worst:.2e}")


def _fast_vs_strict(make, steps):
    """FAST on the fused kernels against STRICT on the stage-by-stage kernels: worst per-component rel L-inf"""
    P, m, dts, info = _run(make(), abi.FP_FAST, True, steps)
    assert info["path"] != 0, "fused kernels"
    fast = [m.download(s) for s in range(m.n_states)]
    m.close()
    P2, r, dts_ref, info_ref = _run(make(), abi.FP_STRICT, False, steps)
    assert info_ref["path"] == 0
    np.testing.assert_allclose(dts, dts_ref, rtol=1e-12)
    worst = 0.0
    for s in range(r.n_states):
        e = rel_linf_per_comp(fast[s], r.download(s))
        assert e.max() <= TOL, (P.state_names[s], e.tolist())
        worst = max(worst, float(e.max()))
    r.close()
    return P, fast, info, worst


def test_kelvin_helmholtz_2048_fast_default_path():
    """BASELINE configs[1] (the pure flux stencil): both directions wrapped inside the marching kernel, no ghost-fill
    launch; mass, momentum and energy conserved to rounding on the periodic domain"""
    P, fast, info, worst = _fast_vs_strict(lambda: problems.kelvin_helmholtz_2d(2048), 6)
    assert info["path"] == 1 and info["wrapped"] == 3, info
    U0 = P.initial_data(0)
    for c in (0, 4):
        a, b = U0[c].sum(dtype=np.longdouble), fast[0][c].sum(dtype=np.longdouble)
        assert abs(b - a) <= 1e-12 * abs(a), (c, float((b - a) / a))
    for c in (1, 2):   # momenta: sums near zero by symmetry, measured against the sum of magnitudes
        a, b = U0[c].sum(dtype=np.longdouble), fast[0][c].sum(dtype=np.longdouble)
        assert abs(b - a) <= 1e-12 * np.abs(U0[c]).sum(dtype=np.longdouble), c
    print(f"kh 2048^2 FAST (bench path) vs STRICT staged: worst per-component rel L-inf {worst:.2e}")


def test_richtmyer_meshkov_256_fast_default_path():
    """BASELINE configs[4] at the size of the bench's reduced leg (256^3; the 512^3 leg runs the same kernels):
    z-marching hydro kernel, three-pass field update with stage C as its epilogue, stage A a launch of its own"""
    P, fast, info, worst = _fast_vs_strict(lambda: problems.richtmyer_meshkov_3d(256), 3)
    assert info["fusion"] == 2, info
    assert np.all(fast[P.state_names.index("field")][8] == 1.0)
    print(f"rmi 256^3 FAST (bench path) vs STRICT staged: worst per-component rel L-inf {worst:.2e}")
