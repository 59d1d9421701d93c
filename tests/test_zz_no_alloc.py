"""cmfp_advance_steps_static — the entry point that allocates nothing (no counterpart in the
reference; for callers that cannot call hipMalloc) — against cmfp_advance_steps through the C ABI.

Two calls in a row share the GPU's one static scratch array (device-resident dt + dt log); the second
call is smaller (smaller block, fewer steps).  Bar: bit-identical states, dt sequences and next dt in
both arithmetic modes (the same kernels run; only the home of the dt words differs).  Collected last
(file name) so that a failure here hides no other test."""
import ctypes as C

import numpy as np
import pytest

from cerberus_amd import abi, config, lib, problems

pytestmark = pytest.mark.gpu


def _pair(cfg, mode):
    from cerberus_amd.mfp import MFP
    out = []
    for _ in range(2):
        m = MFP(config.Problem(cfg, fp_mode=mode))
        m.init_data()
        out.append(m)
    return out


def _steps(m, fn, dt0, n):
    dts, nxt = np.full(n, -1.0), C.c_double(-1.0)
    rc = fn(m.h, 0.0, dt0, n, dts.ctypes.data_as(lib.dp), C.byref(nxt))
    assert rc == 0, (rc, m.L.cmfp_last_error(m.h).decode())
    return dts, nxt.value


def _check(cfg, mode, n):
    a, b = _pair(cfg, mode)
    L = a.L
    dt0 = a.est_time_step()
    assert b.est_time_step() == dt0
    da, na = _steps(a, L.cmfp_advance_steps, dt0, n)
    db, nb = _steps(b, L.cmfp_advance_steps_static, dt0, n)
    assert np.array_equal(da, db) and na == nb and np.all(da > 0)
    assert a.last_advance_path() == b.last_advance_path() != 0
    for s in range(a.n_states):
        assert np.array_equal(a.download(s), b.download(s)), (s, mode)
    # a second batch on the same contexts continues from the returned dt
    da, na2 = _steps(a, L.cmfp_advance_steps, na, 2)
    db, nb2 = _steps(b, L.cmfp_advance_steps_static, nb, 2)
    assert np.array_equal(da, db) and na2 == nb2
    for s in range(a.n_states):
        assert np.array_equal(a.download(s), b.download(s)), (s, mode)
    a.close()
    b.close()


@pytest.mark.parametrize("mode", [abi.FP_STRICT, abi.FP_FAST])
def test_static_matches_allocating_entry_point_twice(mode):
    # first call: two-fluid Orszag-Tang 96^2, 12 steps; second, smaller: 40^2, 5 steps; then other kernel families
    _check(problems.orszag_tang_two_fluid_2d(96), mode, 12)
    _check(problems.orszag_tang_two_fluid_2d(40), mode, 5)
    _check(problems.kelvin_helmholtz_2d(64), mode, 7)
    _check(problems.richtmyer_meshkov_3d(16), mode, 3)


def test_static_rejects_what_it_cannot_do():
    from cerberus_amd.mfp import MFP
    m = MFP(config.Problem(problems.orszag_tang_two_fluid_2d(32), fp_mode=abi.FP_STRICT))
    m.init_data()
    dts, nxt = np.empty(8), C.c_double()
    big = abi.STATIC_MAX_STEPS + 1
    buf = np.empty(big)
    rc = m.L.cmfp_advance_steps_static(m.h, 0.0, 1e-3, big, buf.ctypes.data_as(lib.dp), C.byref(nxt))
    assert rc == abi.ERR_INVALID and b"CMFP_STATIC_MAX_STEPS" in m.L.cmfp_last_error(m.h)
    # a kernel selection whose scratch was not reserved by cmfp_create: refused, nothing allocated, nothing advanced
    before = m.download(0)
    m.L.cmfp_set_fused(m.h, 2)          # gather kernels in 2-D need flux scratch
    rc = m.L.cmfp_advance_steps_static(m.h, 0.0, 1e-3, 4, dts.ctypes.data_as(lib.dp), C.byref(nxt))
    assert rc == abi.ERR_INVALID and b"cmfp_advance" in m.L.cmfp_last_error(m.h)
    assert np.array_equal(m.download(0), before)
    # after one ordinary step the scratch exists and the static entry point runs
    m.advance(1e-3)
    rc = m.L.cmfp_advance_steps_static(m.h, 0.0, 1e-3, 4, dts.ctypes.data_as(lib.dp), C.byref(nxt))
    assert rc == 0 and m.last_advance_path() == 2
    m.close()
