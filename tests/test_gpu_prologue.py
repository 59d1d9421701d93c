"""Source stage A as the prologue of the 2-D marching CTU kernels (mfp_source.h; FAST since round 2,
STRICT since round 3: there bit-identical to the oracle).

The prologue applies the Strang source half step (MFP_advance.cpp:146-197) to every cell a CTU kernel loads
instead of rewriting the states in a launch of its own.  Checked here: it is the path a periodic two-fluid 2-D
block takes (cmfp_last_source_fusion), its results stay within the north-star bound of the oracle, they agree
with the separate launch to rounding, and blocks / option sets outside its cover keep the separate launch."""
import os

import numpy as np
import pytest

from cerberus_amd import abi, config, problems
from helpers import rel_linf

pytestmark = pytest.mark.gpu

TOL_FAST = 1e-12      # north_star: relative L-inf on conserved variables after N steps
# prologue against the separate stage-A launch: the same arithmetic up to operation order, i.e. a few ulp of the
# state's scale per component (measured 1e-17 ... 4e-16 of max|b|, profiles/r03/comp_scales.json); on a component at
# the rounding floor of helpers.rel_linf_per_comp (1e-3 of the state's scale) that reads 4e-13
TOL_SAME = 5e-13


def _gpu(problem, prologue=True, **kw):
    from cerberus_amd.mfp import MFP
    old = os.environ.get("CMFP_SRC_PROLOGUE")
    os.environ["CMFP_SRC_PROLOGUE"] = "1" if prologue else "0"   # read by cmfp_create
    try:
        m = MFP(problem, **kw)
    finally:
        if old is None:
            del os.environ["CMFP_SRC_PROLOGUE"]
        else:
            os.environ["CMFP_SRC_PROLOGUE"] = old
    m.init_data()
    return m


def _ot(nx, ny, faces="pp", **kw):
    """two-fluid Orszag-Tang on nx x ny cells; faces: per direction p(eriodic), o(utflow: foextrap ghost cells) or
    w(all: slipwall species, conducting-wall field = reflect_even / reflect_odd ghost cells)"""
    cfg = problems.orszag_tang_two_fluid_2d(nx, **kw)
    cfg["n_cell"] = [nx, ny]
    wall = {k: {"fill_hydro_bc": "slipwall"} for k in ("lo", "hi")}
    fwall = {k: {"fill_D_bc": "asymmetry", "fill_B_bc": "asymmetry", "fill_phi_bc": "symmetry", "fill_psi_bc": "symmetry",
                 "fill_mu_bc": "symmetry", "fill_ep_bc": "symmetry"} for k in ("lo", "hi")}
    for d, f in enumerate(faces):
        cfg["is_periodic"][d] = int(f == "p")
        if f == "w":
            for name, st in cfg["states"].items():
                st.setdefault("bc", {})["xy"[d]] = fwall if st["type"] == "field" else wall
    return cfg


@pytest.mark.parametrize("shape,steps,faces", [((48, 48), 8, "pp"), ((130, 70), 6, "pp"), ((256, 256), 20, "pp"), ((61, 9), 4, "pp"),
                                               ((130, 70), 6, "pw"), ((70, 130), 6, "wp"), ((96, 64), 6, "ww"), ((130, 70), 6, "op"),
                                               ((96, 64), 6, "wo")])
@pytest.mark.parametrize("source", ["plasma5", "Lorentz"])
def test_prologue_matches_oracle_and_separate_launch(oracle, shape, steps, source, faces):
    cfg = _ot(*shape, faces=faces, source=source)
    P = config.Problem(cfg, fp_mode=abi.FP_FAST)
    ref = oracle.OracleSim(P, max_grid_size=32)
    dts_ref = ref.run(steps)
    a, b = _gpu(P, prologue=True), _gpu(P, prologue=False)
    da, db = a.run(steps), b.run(steps)
    assert a.last_advance_path() == 1 and b.last_advance_path() == 1
    assert a.last_source_fusion() == 3, "stage A and stage C inside the CTU kernels"
    assert b.last_source_fusion() == 2, "CMFP_SRC_PROLOGUE=0 keeps stage A a launch of its own"
    np.testing.assert_allclose(da, dts_ref, rtol=1e-12)
    np.testing.assert_allclose(da, db, rtol=1e-13)
    for s in range(a.n_states):
        ua, ub, ur = a.download(s), b.download(s), ref.get_state(s)
        assert rel_linf(ua, ur) <= TOL_FAST, (P.state_names[s], rel_linf(ua, ur))
        assert rel_linf(ua, ub) <= TOL_SAME, (P.state_names[s], rel_linf(ua, ub))
    a.close()
    b.close()


@pytest.mark.parametrize("shape,steps,faces", [((48, 48), 8, "pp"), ((130, 70), 6, "pp"), ((61, 9), 4, "pp"), ((130, 70), 6, "pw"),
                                               ((70, 130), 6, "wp"), ((96, 64), 6, "ww"), ((130, 70), 6, "op"), ((96, 64), 6, "wo")])
@pytest.mark.parametrize("source", ["plasma5", "Lorentz"])
def test_strict_prologue_is_bit_identical(oracle, shape, steps, source, faces):
    """STRICT arithmetic: the CTU kernels apply stage A to what they load (the hydro waves carry the conserved
    stage-A values of two rows for the update) and the step equals the oracle's bit for bit, dt sequence included —
    as does the run that keeps stage A a launch of its own."""
    cfg = _ot(*shape, faces=faces, source=source)
    P = config.Problem(cfg, fp_mode=abi.FP_STRICT)
    ref = oracle.OracleSim(P, max_grid_size=32)
    dts_ref = ref.run(steps)
    a, b = _gpu(P, prologue=True), _gpu(P, prologue=False)
    da, db = a.run(steps), b.run(steps)
    assert a.last_advance_path() == 1 and a.last_source_fusion() == 3
    assert b.last_advance_path() == 1 and b.last_source_fusion() == 2
    assert da == dts_ref and db == dts_ref
    for s in range(a.n_states):
        ur = ref.get_state(s)
        assert np.array_equal(a.download(s), ur), P.state_names[s]
        assert np.array_equal(b.download(s), ur), P.state_names[s]
    a.close()
    b.close()


@pytest.mark.parametrize("mode", [abi.FP_STRICT, abi.FP_FAST], ids=["strict", "fast"])
@pytest.mark.parametrize("faces", ["pp", "pw"])
def test_prologue_with_the_hybrid_flux(oracle, mode, faces):
    """HLLE/HLLC with the pressure-jump detector: the prologue instance of the hydro kernel runs at one wave per SIMD
    (it does not fit two); STRICT bit-identical to the oracle, FAST within the north-star bound"""
    cfg = _ot(130, 70, faces=faces)
    for n in ("ion", "electron"):
        cfg["states"][n]["flux"] = "HLLE/HLLC"
        cfg["states"][n]["shock_detector"] = {"name": "pressure_jump_detector", "threshold": 0.3}
    P = config.Problem(cfg, fp_mode=mode)
    ref = oracle.OracleSim(P, max_grid_size=32)
    dts_ref = ref.run(6)
    a = _gpu(P)
    da = a.run(6)
    assert a.last_advance_path() == 1 and a.last_source_fusion() == 3
    for s in range(a.n_states):
        if mode == abi.FP_STRICT:
            assert da == dts_ref and np.array_equal(a.download(s), ref.get_state(s)), P.state_names[s]
        else:
            assert rel_linf(a.download(s), ref.get_state(s)) <= TOL_FAST, (P.state_names[s], rel_linf(a.download(s), ref.get_state(s)))
    a.close()


@pytest.mark.parametrize("faces", ["pp", "pw", "wo"])
def test_field_update_by_a_pair_of_waves(oracle, monkeypatch, faces):
    """CMFP_FIELD_PAIR=1: the TE / TM halves of the field update on two waves per strip (k_fld_pair), stages A and C
    alternating between them — the same bits as the oracle (STRICT) with the prologue, with the epilogue only and
    without a source action; FAST the same bits as the one-wave kernel."""
    monkeypatch.setenv("CMFP_FIELD_PAIR", "1")      # read at the library's first field launch: its own process state
    import subprocess, sys, textwrap
    code = textwrap.dedent(f"""
        import os, sys
        sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r}); sys.path.insert(0, {os.path.dirname(os.path.abspath(__file__))!r})
        import numpy as np
        from cerberus_amd import abi, config
        from oracle import oracle
        import test_gpu_prologue as T
        for source in ("plasma5", None):
            for prologue in (True, False):
                cfg = T._ot(130, 70, faces={faces!r})
                if source is None:
                    del cfg["actions"]["plasma"]
                P = config.Problem(cfg, fp_mode=abi.FP_STRICT)
                ref = oracle.OracleSim(P, max_grid_size=32)
                dts_ref = ref.run(5)
                m = T._gpu(P, prologue=prologue)
                assert m.run(5) == dts_ref and m.last_advance_path() == 1
                for s in range(m.n_states):
                    assert np.array_equal(m.download(s), ref.get_state(s)), (source, prologue, P.state_names[s])
                m.close()
        print("pair ok")
    """)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, CMFP_FIELD_PAIR="1"))
    assert out.returncode == 0 and "pair ok" in out.stdout, out.stderr[-2000:]


def test_strict_prologue_with_denormal_components(oracle):
    """components the source does not touch pass through the reference's 0.5 x + 0.5 x (MFP_advance.cpp:194), which
    changes a subnormal value with an odd last bit: the prologue does the same (rho of the species, B of the field)"""
    cfg = _ot(48, 48)
    tiny = 5e-324 * 3          # a subnormal with an odd last bit
    cfg["states"]["field"]["value"]["z_B"] = lambda d: np.where(d["x"] > 3.0, tiny, 0.0)
    P = config.Problem(cfg, fp_mode=abi.FP_STRICT)
    ref = oracle.OracleSim(P)
    ref.run(3)
    a = _gpu(P)
    a.run(3)
    assert a.last_source_fusion() == 3
    for s in range(a.n_states):
        assert np.array_equal(a.download(s), ref.get_state(s)), P.state_names[s]
    a.close()


@pytest.mark.parametrize("faces", ["pp", "pw", "ww"])
@pytest.mark.parametrize("device_dt", [True, False])
def test_prologue_split_launches_and_dt_modes(oracle, device_dt, faces):
    """interior / frame launches (overlap=2 on an unsplit block) and both time-step modes give what the single
    launch gives, bit for bit: the prologue is pointwise on what each launch loads"""
    P = config.Problem(_ot(130, 70, faces=faces), fp_mode=abi.FP_FAST)
    a, b = _gpu(P, overlap=2, device_dt=device_dt), _gpu(P, overlap=0, device_dt=True)
    da, db = a.run(5), b.run(5)
    assert a.last_source_fusion() == 3 and b.last_source_fusion() == 3
    assert da == db
    for s in range(a.n_states):
        assert np.array_equal(a.download(s), b.download(s)), P.state_names[s]
    a.close()
    b.close()


def test_prologue_on_the_gem_block(oracle):
    """BASELINE configs[3]: conducting walls in y (reflect_even / reflect_odd ghost cells on every component)"""
    P = config.Problem(problems.gem_reconnection_2d(128, 64), fp_mode=abi.FP_FAST)
    ref = oracle.OracleSim(P, max_grid_size=32)
    ref.run(10)
    m = _gpu(P)
    m.run(10)
    assert m.last_source_fusion() == 3
    for s in range(m.n_states):
        assert rel_linf(m.download(s), ref.get_state(s)) <= TOL_FAST, P.state_names[s]
    m.close()


def test_prologue_not_taken_outside_its_cover(oracle):
    """the implicit solve, faces whose ghost cells mix mirror images with copies (a slipwall species beside an
    outflow field) and 3-D blocks keep stage A a launch of its own — and their results stay what they were"""
    mixed = _ot(48, 48, faces="pw")
    del mixed["states"]["field"]["bc"]        # the field falls back to outflow (foextrap) at the species' walls
    cases = [
        (mixed, abi.FP_FAST, 4),
        (_ot(48, 48, solver="implicit"), abi.FP_FAST, 3),
        (_ot(48, 48, solver="implicit"), abi.FP_STRICT, 3),
        (problems.richtmyer_meshkov_3d(24, 12, 12), abi.FP_FAST, 3),
    ]
    for cfg, mode, steps in cases:
        P = config.Problem(cfg, fp_mode=mode)
        ref = oracle.OracleSim(P)
        ref.run(steps)
        m = _gpu(P)
        m.run(steps)
        assert m.last_advance_path() != 0
        assert (m.last_source_fusion() & 1) == 0
        for s in range(m.n_states):
            assert rel_linf(m.download(s), ref.get_state(s)) <= TOL_FAST, P.state_names[s]
        m.close()


@pytest.mark.parametrize("shape,faces", [((1100, 900), "pp"), ((2048, 520), "pw"), ((520, 2048), "wp")])
def test_prologue_at_size(shape, faces):
    """blocks with many strips, long segments and several rounds of waves (no oracle at this size): the prologue
    against the separate stage-A launch to rounding, and bit-identical with and without the kernels' own periodic
    wrap and with split launches"""
    P = config.Problem(_ot(*shape, faces=faces), fp_mode=abi.FP_FAST)
    a, b = _gpu(P, prologue=True), _gpu(P, prologue=False)
    da, db = a.run(4), b.run(4)
    assert a.last_source_fusion() == 3 and b.last_source_fusion() == 2
    np.testing.assert_allclose(da, db, rtol=1e-13)
    ua = [a.download(s) for s in range(a.n_states)]
    for s in range(a.n_states):
        assert rel_linf(ua[s], b.download(s)) <= TOL_SAME, (P.state_names[s], rel_linf(ua[s], b.download(s)))
    a.close()
    b.close()
    old = os.environ.get("CMFP_MARCH_WRAP")
    os.environ["CMFP_MARCH_WRAP"] = "0"
    try:
        c = _gpu(P, prologue=True, overlap=2)
    finally:
        if old is None:
            del os.environ["CMFP_MARCH_WRAP"]
        else:
            os.environ["CMFP_MARCH_WRAP"] = old
    dc = c.run(4)
    assert c.last_wrapped_dims() == 0 and dc == da
    for s in range(c.n_states):
        assert np.array_equal(c.download(s), ua[s]), P.state_names[s]
    c.close()
