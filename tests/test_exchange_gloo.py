"""world_size 2, 4 and 8 on CPU (gloo): the block decomposition, neighbour map,
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

NG = 2
NCS = [3, 2]   # two states with different component counts (batched mode packs them back to back)


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
        self._host_t = {"exchange": 0.0, "dt_reduce": 0.0}
        self._exchange_device = "cpu"
        self.p.periodic = periodic
        ny, nx = self.block_n[1], self.block_n[0]
        self.n_states = len(NCS)
        self.As = [np.full((nc, ny + 2 * NG, nx + 2 * NG), np.nan) for nc in NCS]

    def _pack_device(self):
        return torch.device("cpu")

    def _halo_count(self, s, d):
        A = self.As[s]
        other = A.shape[1] if d == 0 else A.shape[2]
        return A.shape[0] * NG * other

    def _slab(self, d, first):
        # NG layers starting at local index `first` (ghost-inclusive offset NG), full extent otherwise
        sl = [slice(None)] * 3
        sl[2 - d] = slice(first + NG, first + 2 * NG)
        return tuple(sl)

    def _halo_pack(self, s, d, side, buf):
        n = self.block_n[d]
        first = 0 if side == 0 else n - NG
        buf.copy_(torch.from_numpy(np.ascontiguousarray(self.As[s][self._slab(d, first)]).ravel()))

    def _halo_unpack(self, s, d, side, buf):
        n = self.block_n[d]
        first = -NG if side == 0 else n
        sl = self._slab(d, first)
        self.As[s][sl] = buf.numpy().reshape(self.As[s][sl].shape)

    def local_wrap(self, d):
        n = self.block_n[d]
        for A in self.As:
            if d == 0:
                A[:, :, :NG] = A[:, :, n:n + NG]
                A[:, :, n + NG:] = A[:, :, NG:2 * NG]
            else:
                A[:, :NG, :] = A[:, n:n + NG, :]
                A[:, n + NG:, :] = A[:, NG:2 * NG, :]


def _worker(rank, world, port, n, periodic, batched, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        P = config.Problem(problems.kelvin_helmholtz_2d(n))
        m = HostHaloMFP(P, rank, world, periodic)
        rng = np.random.default_rng(7)
        Gs = [rng.standard_normal((nc, n, n)) for nc in NCS]      # the global fields, same on every rank
        j0, i0 = m.block_lo[1], m.block_lo[0]
        ny, nx = m.block_n[1], m.block_n[0]
        for A, G in zip(m.As, Gs):
            A[:, NG:-NG, NG:-NG] = G[:, j0:j0 + ny, i0:i0 + nx]
        for d in range(2):                                        # the fill_patch_many loop of the library
            if m.grid[d] == 1:
                if periodic[d]:
                    m.local_wrap(d)
            elif batched:
                assert m._exchange(None, None, -1, d) == 0        # CMFP_ALL_STATES
            else:
                for s in range(m.n_states):
                    assert m._exchange(None, None, s, d) == 0
        # expected: periodic pad of the global field (ghosts at physical faces stay NaN)
        ok = True
        for A, G in zip(m.As, Gs):
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
            ok = ok and np.array_equal(A, exp, equal_nan=True)
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


@pytest.mark.parametrize("batched", [False, True])
@pytest.mark.parametrize("world,periodic", [(2, [1, 1, 0]), (4, [1, 1, 0]), (2, [1, 0, 0]), (4, [0, 1, 0]),
                                            (8, [1, 1, 0]), (8, [1, 0, 0])])   # 8 = the 2 x 4 grid of the N = 8 scaling run
def test_halo_exchange_protocol(world, periodic, batched):
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, 16, periodic, batched, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert all(ret.get(r) for r in range(world)), dict(ret)
