"""world_size 2 and 4 on CPU (gloo): the block decomposition, neighbour map,
message ordering and ghost-slab conventions of MFP._exchange, with the three
device hooks (count / pack / unpack) replaced by a numpy emulation of the
pack kernel's conventions.  Every rank must end up with exactly the ghost cells
a single-block periodic fill gives."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cerberus_amd import config, problems
from cerberus_amd.mfp import MFP, block_grid, rank_coords

NG, NC = 2, 3


class HostHaloMFP(MFP):
    """MFP with the halo hooks on numpy arrays; no device context."""

    def __init__(self, problem, rank, world, periodic):
        self.p = problem
        self.rank, self.world, self.pg = rank, world, None
        self.grid = block_grid(problem.dim, world)
        self.coords = rank_coords(rank, self.grid)
        self.block_n = [problem.n_cell[d] // self.grid[d] for d in range(3)]
        self.block_lo = [self.coords[d] * self.block_n[d] for d in range(3)]
        self._bufs = {}
        self._exchange_device = "cpu"
        self.p.periodic = periodic
        ny, nx = self.block_n[1], self.block_n[0]
        self.A = np.full((NC, ny + 2 * NG, nx + 2 * NG), np.nan)

    def _pack_device(self):
        return torch.device("cpu")

    def _halo_count(self, s, d):
        other = self.A.shape[1] if d == 0 else self.A.shape[2]
        return NC * NG * other

    def _slab(self, d, first):
        # NG layers starting at local index `first` (ghost-inclusive offset NG), full extent otherwise
        sl = [slice(None)] * 3
        sl[2 - d] = slice(first + NG, first + 2 * NG)
        return tuple(sl)

    def _halo_pack(self, s, d, side, buf):
        n = self.block_n[d]
        first = 0 if side == 0 else n - NG
        buf.copy_(torch.from_numpy(np.ascontiguousarray(self.A[self._slab(d, first)]).ravel()))

    def _halo_unpack(self, s, d, side, buf):
        n = self.block_n[d]
        first = -NG if side == 0 else n
        sl = self._slab(d, first)
        self.A[sl] = buf.numpy().reshape(self.A[sl].shape)

    def local_wrap(self, d):
        n = self.block_n[d]
        if d == 0:
            self.A[:, :, :NG] = self.A[:, :, n:n + NG]
            self.A[:, :, n + NG:] = self.A[:, :, NG:2 * NG]
        else:
            self.A[:, :NG, :] = self.A[:, n:n + NG, :]
            self.A[:, n + NG:, :] = self.A[:, NG:2 * NG, :]


def _worker(rank, world, port, n, periodic, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        P = config.Problem(problems.kelvin_helmholtz_2d(n))
        m = HostHaloMFP(P, rank, world, periodic)
        rng = np.random.default_rng(7)
        G = rng.standard_normal((NC, n, n))                       # the global field, same on every rank
        j0, i0 = m.block_lo[1], m.block_lo[0]
        ny, nx = m.block_n[1], m.block_n[0]
        m.A[:, NG:-NG, NG:-NG] = G[:, j0:j0 + ny, i0:i0 + nx]
        for d in range(2):                                        # the fill_patch loop of the library
            if m.grid[d] == 1:
                if periodic[d]:
                    m.local_wrap(d)
            else:
                assert m._exchange(None, None, 0, d) == 0
        # expected: periodic pad of the global field (ghosts at physical faces stay NaN)
        pad = np.pad(G, ((0, 0), (NG, NG), (NG, NG)), mode="wrap")
        exp = pad[:, j0:j0 + ny + 2 * NG, i0:i0 + nx + 2 * NG].copy()
        for d in range(2):
            if not periodic[d]:
                ax = 2 - d
                if m.coords[d] == 0:
                    sl = [slice(None)] * 3
                    sl[ax] = slice(0, NG)
                    exp[tuple(sl)] = np.nan
                if m.coords[d] == m.grid[d] - 1:
                    sl = [slice(None)] * 3
                    sl[ax] = slice(-NG, None)
                    exp[tuple(sl)] = np.nan
        ok = np.array_equal(m.A, exp, equal_nan=True)
        # dt min-reduction path
        m._exchange_device = "cpu"
        t = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        ret[rank] = bool(ok) and float(t.item()) == 1.0
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,periodic", [(2, [1, 1, 0]), (4, [1, 1, 0]), (2, [1, 0, 0]), (4, [0, 1, 0])])
def test_halo_exchange_protocol(world, periodic):
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, 16, periodic, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert all(ret.get(r) for r in range(world)), dict(ret)
