"""Leaf numerics on the GPU through the C ABI: the survey's known-answer
vectors (outputs of the reference's own leaf sources) and random inputs against
the CPU oracle.  STRICT mode must be bit-exact, FAST (FMA) within 1e-13."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from cerberus_amd import abi, lib

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "survey_kat.json")))


@pytest.fixture(scope="module")
def dev():
    import torch
    assert torch.cuda.is_available()
    return torch


def _t(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda()


def gpu_reconstruct(torch, recon, st, mode):
    L = lib.load()
    n = st.shape[1]
    d = _t(torch, st)
    out = [torch.empty(n, dtype=torch.float64, device="cuda") for _ in range(3)]
    rc = L.cmfp_leaf_reconstruct(recon, mode, n, d.data_ptr(), out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), None)
    assert rc == 0
    torch.cuda.synchronize()
    return [o.cpu().numpy() for o in out]


def gpu_riemann(torch, flux, Lq, Rq, mode, c0=0.0, ch=0.0, thr=-1.0):
    L = lib.load()
    n = Lq.shape[1]
    a, b = _t(torch, Lq), _t(torch, Rq)
    F = torch.empty((10 if flux == 10 else 5, n), dtype=torch.float64, device="cuda")
    assert L.cmfp_leaf_riemann(flux, mode, n, a.data_ptr(), b.data_ptr(), F.data_ptr(), c0, ch, thr, None) == 0
    torch.cuda.synchronize()
    return F.cpu().numpy()


col = lambda v: np.array(v, dtype=float)[:, None]


def test_kat_on_gpu(dev):
    for mode in (abi.FP_STRICT, abi.FP_FAST):
        tol = 0.0 if mode == abi.FP_STRICT else 1e-14
        F = gpu_riemann(dev, 0, col(KAT["hydro_L"]), col(KAT["hydro_R"]), mode)[:, 0]
        assert np.abs(F - KAT["HLLC_F"]).max() <= tol * 4
        F = gpu_riemann(dev, 1, col(KAT["hydro_L"]), col(KAT["hydro_R"]), mode)[:, 0]
        assert np.abs(F - KAT["HLLE_F"]).max() <= tol * 4
        F = gpu_riemann(dev, 10, col(KAT["field_L"]), col(KAT["field_R"]), mode, c0=20.0, ch=20.0)[:, 0]
        assert np.abs(F - KAT["field_F"]).max() <= tol * 20
        for name, r, key in [("minmod", 1, "stencil3"), ("vanLeer", 2, "stencil3"), ("MC", 3, "stencil3"),
                             ("O6", 5, "stencil7"), ("WENO", 6, "stencil7")]:
            s, lo, hi = gpu_reconstruct(dev, r, col(KAT[key]), mode)
            assert abs(lo[0] - KAT[name][0]) <= tol and abs(hi[0] - KAT[name][1]) <= tol, (name, mode)


def random_stencils(rng, sl, n):
    """sign changes, exact zeros, equal neighbours, magnitudes 1e-12 .. 1e6 (SURVEY §8c fixture set 1)"""
    st = rng.standard_normal((sl, n)) * 10.0 ** rng.integers(-12, 7, size=(1, n))
    st[:, : n // 10] = np.round(st[:, : n // 10])            # ties and zeros
    if sl > 1:
        st[1, n // 10: n // 5] = st[0, n // 10: n // 5]      # equal neighbours
    st[:, n // 5: n // 4] = 0.0
    return st


@pytest.mark.parametrize("recon", range(7))
def test_reconstruction_random(dev, oracle, recon):
    rng = np.random.default_rng(recon)
    st = random_stencils(rng, abi.RECON_STENCIL[recon], 20000)
    ref = oracle.reconstruct(recon, st)
    got = gpu_reconstruct(dev, recon, st, abi.FP_STRICT)
    for a, b in zip(got, ref):
        assert np.array_equal(a, b)
    got = gpu_reconstruct(dev, recon, st, abi.FP_FAST)
    for a, b in zip(got, ref):
        scale = np.abs(st).max(axis=0) + 1e-300
        assert (np.abs(a - b) / scale).max() < 1e-12


def random_hydro_states(rng, n):
    """all four HLLC branches, p* above and below p_K, electron-like densities"""
    def side():
        q = np.empty((8, n))
        q[0] = 10.0 ** rng.uniform(-3, 1, n)
        q[1:4] = rng.standard_normal((3, n)) * 10.0 ** rng.uniform(-1, 1, (1, n))
        q[4] = 10.0 ** rng.uniform(-3, 1, n)
        q[6] = 5.0 / 3.0
        q[5] = q[4] / q[0]
        q[7] = 2.5
        return q
    Lq, Rq = side(), side()
    Lq[1, : n // 4] += 30.0      # supersonic to the right: S_L >= 0
    Rq[1, : n // 4] += 30.0
    Lq[1, n // 4: n // 2] -= 30.0  # supersonic to the left: S_R < 0
    Rq[1, n // 4: n // 2] -= 30.0
    return Lq, Rq


@pytest.mark.parametrize("flux,thr", [(0, -1.0), (1, -1.0), (2, 0.4)])
def test_hydro_riemann_random(dev, oracle, flux, thr):
    rng = np.random.default_rng(10 + flux)
    Lq, Rq = random_hydro_states(rng, 40000)
    ref = oracle.riemann(flux, Lq, Rq, shock_threshold=thr)
    got = gpu_riemann(dev, flux, Lq, Rq, abi.FP_STRICT, thr=thr)
    assert np.array_equal(got, ref)
    got = gpu_riemann(dev, flux, Lq, Rq, abi.FP_FAST, thr=thr)
    scale = np.abs(ref).max(axis=0) + 1e-300
    assert (np.abs(got - ref) / scale).max() < 1e-11       # cancellation in the star-state formulas


@pytest.mark.parametrize("ch", [0.0, 20.0, 7.5])
def test_field_riemann_random(dev, oracle, ch):
    rng = np.random.default_rng(3)
    n = 20000
    Lq, Rq = rng.standard_normal((10, n)), rng.standard_normal((10, n))
    for q in (Lq, Rq):
        q[8] = rng.uniform(0.5, 2.0, n)     # non-unit mu, ep per side
        q[9] = rng.uniform(0.5, 2.0, n)
    ref = oracle.riemann(10, Lq, Rq, c0=20.0, ch=ch)
    got = gpu_riemann(dev, 10, Lq, Rq, abi.FP_STRICT, c0=20.0, ch=ch)
    assert np.array_equal(got, ref)
    got = gpu_riemann(dev, 10, Lq, Rq, abi.FP_FAST, c0=20.0, ch=ch)
    assert np.abs(got - ref).max() < 1e-11 * np.abs(ref).max()


@pytest.mark.parametrize("mass,gamma", [(1.0, 5 / 3), (0.04, 5 / 3), (0.01, 1.4)])
def test_gas_random(dev, oracle, mass, gamma):
    torch = dev
    L = lib.load()
    rng = np.random.default_rng(5)
    n = 20000
    U = np.empty((5, n))
    U[0] = 10.0 ** rng.uniform(-3, 1, n)
    U[1:4] = rng.standard_normal((3, n)) * U[0]
    U[4] = 0.5 * (U[1:4] ** 2).sum(axis=0) / U[0] + 10.0 ** rng.uniform(-3, 1, n)
    Qr, sr = oracle.cons2prim(mass, gamma, U)
    for mode in (abi.FP_STRICT, abi.FP_FAST):
        d = _t(torch, U)
        Q = torch.empty((8, n), dtype=torch.float64, device="cuda")
        s = torch.empty((3, n), dtype=torch.float64, device="cuda")
        assert L.cmfp_leaf_cons2prim(mode, n, mass, gamma, d.data_ptr(), Q.data_ptr(), s.data_ptr(), None) == 0
        U2 = torch.empty((5, n), dtype=torch.float64, device="cuda")
        assert L.cmfp_leaf_prim2cons(mode, n, mass, gamma, Q.data_ptr(), U2.data_ptr(), None) == 0
        torch.cuda.synchronize()
        Ur = oracle.prim2cons(mass, gamma, Qr)
        if mode == abi.FP_STRICT:
            assert np.array_equal(Q.cpu().numpy(), Qr) and np.array_equal(s.cpu().numpy(), sr)
            assert np.array_equal(U2.cpu().numpy(), Ur)
        else:
            np.testing.assert_allclose(Q.cpu().numpy(), Qr, rtol=1e-11, atol=1e-13)
            np.testing.assert_allclose(s.cpu().numpy(), sr, rtol=1e-11)
