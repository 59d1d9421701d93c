"""Periodic directions wrapped inside the 2-D marching kernels (no ghost-fill launch for them).

A block that spans a periodic direction alone has that direction's ghost cells filled by a copy of its own far
side (FillPatch, MFP_action.cpp:59).  Where every CTU state of the step runs on a marching kernel the kernels read
the far side directly instead.  Checked: the step takes that path (cmfp_last_wrapped_dims), its results are
bit-identical to the step with the fill launches (STRICT and FAST), STRICT stays bit-identical to the oracle, and
blocks / kernels outside its cover keep the fill."""
import os

import numpy as np
import pytest

from cerberus_amd import abi, config, problems

pytestmark = pytest.mark.gpu


def _gpu(problem, wrap=True, **kw):
    """wrap: True / False = wrap where it applies / never; 2 / 4 = the bit form of CMFP_MARCH_WRAP: x only / y only"""
    from cerberus_amd.mfp import MFP
    old = os.environ.get("CMFP_MARCH_WRAP")
    os.environ["CMFP_MARCH_WRAP"] = str(int(wrap))   # read by cmfp_create
    try:
        m = MFP(problem, **kw)
    finally:
        if old is None:
            del os.environ["CMFP_MARCH_WRAP"]
        else:
            os.environ["CMFP_MARCH_WRAP"] = old
    m.init_data()
    return m


def _shape(cfg, nx, ny):
    cfg["n_cell"] = [nx, ny]
    return cfg


CASES = [
    ("kh", lambda: _shape(problems.kelvin_helmholtz_2d(64), 130, 70), 3),
    ("kh_narrow", lambda: _shape(problems.kelvin_helmholtz_2d(64), 6, 64), 3),
    ("ot", lambda: _shape(problems.orszag_tang_two_fluid_2d(64), 130, 70), 3),
    ("ot_narrow", lambda: _shape(problems.orszag_tang_two_fluid_2d(64), 61, 6), 3),
    ("ot_implicit", lambda: problems.orszag_tang_two_fluid_2d(48, solver="implicit"), 3),
    ("gem", lambda: problems.gem_reconnection_2d(96, 48), 1),      # periodic in x, conducting walls in y
]


@pytest.mark.parametrize("mode", [abi.FP_STRICT, abi.FP_FAST], ids=["strict", "fast"])
@pytest.mark.parametrize("overlap", [1, 2])
@pytest.mark.parametrize("name,make,dims", CASES, ids=[c[0] for c in CASES])
def test_wrapped_step_equals_filled_step(oracle, name, make, dims, overlap, mode):
    P = config.Problem(make(), fp_mode=mode)
    a, b = _gpu(P, wrap=True, overlap=overlap), _gpu(P, wrap=False, overlap=overlap)
    da, db = a.run(6), b.run(6)
    assert a.last_advance_path() == 1 and b.last_advance_path() == 1
    if name == "ot_implicit" and overlap == 2:
        dims = 0    # no source epilogue with the implicit solve: the field's interior / frame launches are the LDS-tiled kernel's
    assert a.last_wrapped_dims() == dims and b.last_wrapped_dims() == 0
    assert da == db
    for s in range(a.n_states):
        assert np.array_equal(a.download(s), b.download(s)), (name, P.state_names[s])
    if mode == abi.FP_STRICT and overlap == 1 and name != "ot_implicit":   # (the implicit solve is within 1e-12 of the oracle's QR, not bit-identical)
        ref = oracle.OracleSim(P, max_grid_size=32)
        assert ref.run(6) == da
        for s in range(a.n_states):
            assert np.array_equal(a.download(s), ref.get_state(s)), (name, P.state_names[s])
    a.close()
    b.close()


@pytest.mark.parametrize("mode", [abi.FP_STRICT, abi.FP_FAST], ids=["strict", "fast"])
@pytest.mark.parametrize("wrap,dims", [(4, 2), (2, 1)], ids=["y_only", "x_only"])
@pytest.mark.parametrize("name", ["ot", "kh"])
def test_one_direction_wrapped_the_other_filled(oracle, name, wrap, dims, mode):
    """What a 4096^2 step does (x ghost-fill launch + y wrapped in the kernels, the source prologue on top in FAST:
    mfp_context.cpp `wrap_flag`), at a size the oracle reaches: STRICT bit-identical to the oracle, FAST within the
    north-star bound per conserved component, both bit-identical to the step with both fills."""
    from helpers import rel_linf
    make = dict((c[0], c[1]) for c in CASES)[name]
    P = config.Problem(make(), fp_mode=mode)
    a, b = _gpu(P, wrap=wrap), _gpu(P, wrap=False)
    da, db = a.run(6), b.run(6)
    assert a.last_advance_path() == 1 and a.last_wrapped_dims() == dims and b.last_wrapped_dims() == 0
    if name == "ot" and mode == abi.FP_FAST:
        assert a.last_source_fusion() == 3
    assert da == db
    ref = oracle.OracleSim(P, max_grid_size=32)
    dr = ref.run(6)
    for s in range(a.n_states):
        ua = a.download(s)
        assert np.array_equal(ua, b.download(s)), (name, P.state_names[s])
        if mode == abi.FP_STRICT:
            assert da == dr and np.array_equal(ua, ref.get_state(s)), (name, P.state_names[s])
        else:
            np.testing.assert_allclose(da, dr, rtol=1e-12)
            assert rel_linf(ua, ref.get_state(s)) <= 1e-12, (name, P.state_names[s], rel_linf(ua, ref.get_state(s)))
    a.close()
    b.close()


def test_fill_patch_on_request_after_wrapped_steps(oracle):
    """the ghost cells of a wrapped direction are stale after a step; cmfp_fill_patch fills them as before"""
    P = config.Problem(problems.kelvin_helmholtz_2d(48), fp_mode=abi.FP_STRICT)
    a, b = _gpu(P, wrap=True), _gpu(P, wrap=False)
    a.run(4)
    b.run(4)
    assert a.last_wrapped_dims() == 3
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    ghosted = []
    for m in (a, b):
        m.fill_patch(0)
        ptr, ng, ext, nc = m.device_view(0)
        host = np.empty(nc * ext[0] * ext[1] * ext[2])
        assert hip.hipDeviceSynchronize() == 0
        assert hip.hipMemcpy(host.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), C.c_size_t(8 * host.size), 2) == 0
        gs = m.L.cmfp_state_ngrow(m.h, 0)            # the layers FillPatch fills: the state's own num_grow
        A = host.reshape(nc, ext[2], ext[1], ext[0])
        ghosted.append(A[:, :, ng - gs:ext[1] - (ng - gs), ng - gs:ext[0] - (ng - gs)].copy())
    assert np.array_equal(ghosted[0], ghosted[1])
    a.close()
    b.close()


def test_wrap_not_taken_outside_its_cover():
    """the LDS-tiled kernels (fused=3), the gather kernels (fused=2), 3-D blocks and non-periodic blocks keep the fill"""
    for cfg, kw in [(problems.kelvin_helmholtz_2d(48), {"fused": 3}), (problems.kelvin_helmholtz_2d(48), {"fused": 2}),
                    (problems.richtmyer_meshkov_3d(24, 12, 12), {})]:
        m = _gpu(config.Problem(cfg, fp_mode=abi.FP_FAST), **kw)
        m.run(2)
        assert m.last_advance_path() != 0 and m.last_wrapped_dims() == 0
        m.close()
