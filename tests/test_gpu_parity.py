"""GPU parity: the HIP path through the C ABI against the CPU oracle."""
import numpy as np
import pytest

from cerberus_amd import abi, config, problems
from helpers import rel_linf, small_cases, survey_cases

pytestmark = pytest.mark.gpu

TOL_FAST = 1e-12   # north_star: relative L-inf on conserved variables after N steps
TOL_STRICT = 1e-13  # no-FMA build: expected bit-identical except the implicit solve; see test bodies


def _gpu(problem, **kw):
    from cerberus_amd.mfp import MFP
    m = MFP(problem, **kw)
    m.init_data()
    return m


@pytest.mark.parametrize("fused", [False, True, 2, 3])   # staged / fused (2-D: marching hydro + tiled field) / gather / tiled kernels only
@pytest.mark.parametrize("case", range(5))
def test_nstep_parity_strict(oracle, case, fused):
    name, cfg, steps = small_cases()[case]
    P = config.Problem(cfg, fp_mode=abi.FP_STRICT)
    ref = oracle.OracleSim(P)
    dts_ref = ref.run(steps)
    m = _gpu(P, fused=fused)
    dts = m.run(steps)
    assert dts == dts_ref, (name, dts, dts_ref)   # bit-identical time steps
    for s in range(m.n_states):
        a, b = m.download(s), ref.get_state(s)
        assert np.array_equal(a, b), (name, P.state_names[s], rel_linf(a, b))
    m.close()


@pytest.mark.parametrize("fused", [False, True, 2, 3])   # staged / fused (2-D: marching hydro + tiled field) / gather / tiled kernels only
@pytest.mark.parametrize("case", range(5))
def test_nstep_parity_fast(oracle, case, fused):
    name, cfg, steps = small_cases()[case]
    P = config.Problem(cfg, fp_mode=abi.FP_FAST)
    ref = oracle.OracleSim(P)
    dts_ref = ref.run(steps)
    m = _gpu(P, fused=fused)
    dts = m.run(steps)
    np.testing.assert_allclose(dts, dts_ref, rtol=1e-12)
    for s in range(m.n_states):
        a, b = m.download(s), ref.get_state(s)
        assert rel_linf(a, b) <= TOL_FAST, (name, P.state_names[s], rel_linf(a, b))
    m.close()


@pytest.mark.parametrize("fused", [True, 3])
@pytest.mark.parametrize("case", [1, 2, 3])
def test_split_launches_identical(oracle, case, fused):
    """overlap=2: interior tiles on the second stream + frame of boundary tiles after the fill,
    on an unsplit block; must equal the oracle bit for bit like the single launch."""
    name, cfg, steps = small_cases()[case]
    P = config.Problem(cfg, fp_mode=abi.FP_STRICT)
    ref = oracle.OracleSim(P)
    dts_ref = ref.run(steps)
    m = _gpu(P, overlap=2, fused=fused)
    dts = m.run(steps)
    assert m.last_advance_path() == 1
    assert dts == dts_ref
    for s in range(m.n_states):
        assert np.array_equal(m.download(s), ref.get_state(s)), (name, P.state_names[s])
    m.close()


@pytest.mark.parametrize("case", range(5))
def test_device_resident_dt_equals_per_step_calls(case):
    """cmfp_advance_steps (dt kept on the device, estTimeStep + computeNewDt in a kernel) against the
    per-step cmfp_advance / cmfp_est_time_step / cmfp_compute_new_dt loop: identical dt sequence and
    identical states (STRICT), in 1-D, 2-D and 3-D, also when called twice in a row."""
    name, cfg, steps = small_cases()[case]
    P = config.Problem(cfg, fp_mode=abi.FP_STRICT)
    a, b = _gpu(P, device_dt=True), _gpu(P, device_dt=False)
    da = a.run(3) + a.run(steps)
    db = b.run(3) + b.run(steps)
    assert a._device_dt and not b._device_dt
    assert da == db and a.dt == b.dt and a.time == b.time and a.step == b.step
    for s in range(a.n_states):
        assert np.array_equal(a.download(s), b.download(s)), (name, P.state_names[s])
    a.close()
    b.close()


@pytest.mark.parametrize("case", range(5))
def test_survey_size_parity(oracle, case):
    """The parity runs at the sizes SURVEY.md section 8(d) names (2048 cells x 100 steps; 256^2 x 20; 256 x 128 x 20;
    64^3 x 10) on the default (fused) kernels: STRICT bit-identical states and dt sequence, FAST - the arithmetic
    bench.py measures - within the north-star bound of 1e-12 relative L-infinity after all N steps."""
    name, cfg, steps = survey_cases()[case]
    P = config.Problem(cfg, fp_mode=abi.FP_STRICT)
    ref = oracle.OracleSim(P, max_grid_size=32)
    dts_ref = ref.run(steps)
    refs = [ref.get_state(s) for s in range(len(P.state_names))]
    del ref
    m = _gpu(P)
    dts = m.run(steps)
    assert m.last_advance_path() != 0
    assert dts == dts_ref, name
    for s in range(m.n_states):
        assert np.array_equal(m.download(s), refs[s]), (name, P.state_names[s], rel_linf(m.download(s), refs[s]))
    m.close()
    m = _gpu(config.Problem(cfg, fp_mode=abi.FP_FAST))
    dts = m.run(steps)
    np.testing.assert_allclose(dts, dts_ref, rtol=1e-12)
    for s in range(m.n_states):
        assert rel_linf(m.download(s), refs[s]) <= TOL_FAST, (name, P.state_names[s], rel_linf(m.download(s), refs[s]))
    m.close()


ALL_GOLDEN = small_cases() + survey_cases()


@pytest.mark.gpu
@pytest.mark.parametrize("name,cfg,steps", ALL_GOLDEN, ids=[c[0] for c in ALL_GOLDEN])
def test_strict_gpu_matches_committed_nstep_vectors(name, cfg, steps):
    """STRICT HIP path against the committed N-step vectors (tests/golden/oracle_nstep.json): dt sequence
    and every conserved value bit for bit, without the oracle in the loop"""
    import hashlib
    import json
    import os
    G = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_nstep.json")))[name]
    P = config.Problem(cfg, fp_mode=abi.FP_STRICT)
    m = _gpu(P)
    dts = m.run(G["steps"])
    assert [float(v).hex() for v in dts] == G["dts"]
    for s, sname in enumerate(P.state_names):
        a = np.ascontiguousarray(m.download(s), dtype=np.float64)
        assert list(a.shape) == G["states"][sname]["shape"]
        assert hashlib.sha256(a.tobytes()).hexdigest() == G["states"][sname]["sha256"], (name, sname)
    m.close()
