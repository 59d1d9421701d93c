"""Host-side level driver over the C ABI: the role of `MFP : AmrLevel` plus the
`Amr::coarseTimeStep` loop (Source/MFP.H:27-255, MFP_advance.cpp,
MFP_timestep.cpp, main.cpp:154-162) for one uniform level.

One process per GPU.  With world_size > 1 the index space is cut into a
Cartesian grid of blocks (one per rank); ghost faces travel between ranks as
packed device buffers through torch.distributed point-to-point ops (RCCL on
GPU, gloo in the CPU tests of the exchange logic) and dt is min-reduced.  torch
is used for device buffers, streams and the process group only.
"""
import ctypes as C

import numpy as np

from . import abi, lib
from .config import Problem


def block_grid(dim, world):
    """Cartesian process grid: 2-D -> 1x2, 2x2, 2x4; 3-D -> 1x1x2, 1x2x2, 2x2x2
    (SURVEY.md §8e).  The last dimension is split first."""
    grid = [1, 1, 1]
    d = dim - 1
    w = world
    while w > 1:
        if w % 2:
            raise ValueError("world size must be a power of two")
        grid[d] *= 2
        w //= 2
        d = d - 1 if d > 0 else dim - 1
    return grid


def rank_coords(rank, grid):
    return [rank % grid[0], (rank // grid[0]) % grid[1], rank // (grid[0] * grid[1])]


def coords_rank(c, grid):
    return c[0] + grid[0] * (c[1] + grid[1] * c[2])


class MFP:
    def __init__(self, problem, device=0, rank=0, world=1, stream=0, fused=True, pg=None, exchange_device=None,
                 batched_exchange=True, overlap=1, device_dt=True, transport="callback"):
        """transport (world > 1): "callback" = halos and the dt word through torch.distributed from the
        exchange / dt-reduce callbacks (RCCL device buffers, or gloo with host staging where exchange_device is
        "cpu"); "peer" = the library's device-side transport (cmfp_peer_*: neighbours write into each other's
        IPC-mapped arenas, no host code per step); torch.distributed then only carries the set-up (the handles)."""
        self.L = lib.load()
        self.p = problem if isinstance(problem, Problem) else Problem(problem)
        p = self.p
        self.rank, self.world, self.pg = rank, world, pg
        self._stream = int(stream or 0)
        self._host_t = {"exchange": 0.0, "dt_reduce": 0.0}
        self.grid = block_grid(p.dim, world)
        self.coords = rank_coords(rank, self.grid)
        self.block_n, self.block_lo = [1, 1, 1], [0, 0, 0]
        for d in range(3):
            if p.n_cell[d] % self.grid[d]:
                raise ValueError("n_cell must be divisible by the process grid")
            self.block_n[d] = p.n_cell[d] // self.grid[d]
            self.block_lo[d] = self.coords[d] * self.block_n[d]
        self._geom = p.geometry(self.block_lo, self.block_n)
        self._params = p.params()
        self._sd, self._ad = p.state_descs(), p.action_descs()
        self.n_states, self.n_actions = len(p.state_names), len(p.action_names)
        h = C.c_void_p()
        rc = self.L.cmfp_create(C.byref(self._geom), C.byref(self._params), self.n_states, self._sd, self.n_actions,
                                self._ad, device, C.c_void_p(stream), C.byref(h))
        self.h = h
        if rc:
            msg = self.L.cmfp_last_error(h).decode() if h else "no context"
            if rc == abi.ERR_NODEVICE:
                msg = "no HIP device visible: the hyperbolic update has no CPU fallback"
            raise lib.CmfpError(rc, msg)
        self.L.cmfp_set_fused(self.h, int(fused))
        self._chk(self.L.cmfp_set_overlap(self.h, int(overlap)))
        self.shape = tuple(self.block_n[::-1])
        self.time, self.dt, self.step = 0.0, None, 0
        self._elliptic = p.elliptic_actions()
        # with an apply_change action the step is not one enqueue-able unit: per-step driver
        self._device_dt = bool(device_dt) and not self._elliptic
        self._time_bc = [s for s in range(self.n_states) if p.face_values_depend_on_time(s)]
        if self._time_bc:
            self._device_dt = False      # the face tables are re-tabulated before every step: per-step driver
        if self._time_bc and p.scheme != "strang":
            raise NotImplementedError("time-dependent boundary functions: 'strang' only (one flux evaluation at t + dt)")
        for s in range(self.n_states):
            self._upload_face_values(s, 0.0)
        # 'dynamic' components (field states): tabulated at the time the fluxes are evaluated at, before every step
        # when they depend on time, else once; such a state runs stage by stage, step by step
        self._dynamic = [s for s in range(self.n_states) if p.dynamic_entries(s)]
        self._dynamic_t = [s for s in self._dynamic if p.dynamic_depends_on_time(s)]
        if self._dynamic:
            self._device_dt = False
        for s in self._dynamic:
            self._upload_dynamic(s, 0.0)
        self._cb = None
        self._bufs = {}
        self._exchange_device = exchange_device
        self.transport = transport if world > 1 else "none"
        if world > 1 and transport == "peer":
            self._peer_setup()
        elif world > 1:
            self._cb = lib.EXCHANGE_FN(self._exchange)
            self._chk(self.L.cmfp_set_exchange_callback(self.h, self._cb, None))
            self._chk(self.L.cmfp_set_exchange_batched(self.h, int(batched_exchange)))
            if self._device_dt:
                # cmfp_advance_steps on split blocks: the cross-rank dt minimum stays on the device
                self._dt_cb = lib.DT_REDUCE_FN(self._dt_reduce)
                self._dt_buf = None
                self._chk(self.L.cmfp_set_dt_reduce_callback(self.h, self._dt_cb, None))

    # ------------------------------------------------------------------
    def _chk(self, rc):
        if rc:
            raise lib.CmfpError(rc, self.L.cmfp_last_error(self.h).decode())

    def close(self):
        if getattr(self, "_proj", None) is not None:
            self._proj.close()
            self._proj = None
        if getattr(self, "h", None):
            self.L.cmfp_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- data -------------------------------------------------------------
    def init_data(self):
        """MFP::initData (MFP.cpp:156): evaluate the initial conditions on this block."""
        for s in range(self.n_states):
            self.upload(s, self.p.initial_data(s, self.block_lo, self.block_n))

    def upload(self, s, data):
        data = np.ascontiguousarray(data, dtype=np.float64)
        assert data.shape == (self.p.n_comp(s),) + self.shape, (data.shape, self.shape)
        self._chk(self.L.cmfp_state_upload(self.h, s, data.ctypes.data_as(lib.dp)))

    def download(self, s):
        out = np.empty((self.p.n_comp(s),) + self.shape)
        self._chk(self.L.cmfp_state_download(self.h, s, out.ctypes.data_as(lib.dp)))
        return out

    def write_plotfile(self, path, max_grid_size=None):
        """MFP::writePlotFile + writePlotFilePost (MFP_io.cpp:305-422, 467-515): AMReX plotfile of
        the current new-time data that the reference's vis/get_boxlib.py reads; every rank writes
        its block, rank 0 the headers."""
        from . import plotfile
        gather = None
        if self.world > 1:
            import torch.distributed as dist

            def gather(obj):
                out = [None] * self.world
                dist.all_gather_object(out, obj, group=self.pg)
                return out
        states = [self.download(s) for s in range(self.n_states)]
        return plotfile.write_plotfile(path, self.p, states, self.time, self.step, self.block_lo, self.block_n,
                                       max_grid_size, self.rank, self.world, gather)

    def update_download(self, s, which):
        out = np.empty((self.p.n_comp(s),) + self.shape)
        self._chk(self.L.cmfp_update_download(self.h, s, which, out.ctypes.data_as(lib.dp)))
        return out

    def device_view(self, s):
        ptr, ng, ext, nc = C.c_void_p(), C.c_int(), (C.c_longlong * 3)(), C.c_int()
        self._chk(self.L.cmfp_state_device_view(self.h, s, C.byref(ptr), C.byref(ng), C.byref(ext), C.byref(nc)))
        return ptr.value, ng.value, list(ext), nc.value

    def _upload_face_values(self, s, t):
        """bc.<dim>.<side>.<primitive> values / functions of (x, y, z, t), tabulated on this block."""
        if not self.p.face_entries(s):
            return
        ng = self.device_view(s)[1]
        for d, side, prim, cell, face in self.p.face_tables(s, ng, t, self.block_lo, self.block_n):
            self._chk(self.L.cmfp_state_set_face_values(self.h, s, d, side, prim, cell.ctypes.data_as(lib.dp),
                                                         face.ctypes.data_as(lib.dp)))

    def _upload_dynamic(self, s, t):
        ng = self.device_view(s)[1]
        for comp, values in self.p.dynamic_tables(s, ng, t, self.block_lo, self.block_n):
            self._chk(self.L.cmfp_state_set_dynamic(self.h, s, comp, values.ctypes.data_as(lib.dp)))

    # -- stepping ---------------------------------------------------------
    def advance(self, dt):
        for s in self._time_bc:     # the flux stage of the Strang step is evaluated at time + dt (MFP_advance.cpp:208-217)
            self._upload_face_values(s, self.time + dt)
        for s in self._dynamic_t:   # calc_spatial_derivative(time + dt) in the Strang step, (time) in euler / RK2 (:72-104)
            self._upload_dynamic(s, self.time + dt if self.p.scheme == "strang" else self.time)
        self._chk(self.L.cmfp_advance(self.h, self.time, dt))
        for fs, species in self._elliptic:     # act->apply_change (MFP_advance.cpp:49)
            if self.world > 1 and self.p.dim > 1:
                self._project_gathered(fs, species)
                continue
            arr = (C.c_int * max(1, len(species)))(*species)
            bc_B, bc_D = self.p.elliptic_face_types(fs)
            self._chk(self.L.cmfp_elliptic_project_bc(self.h, fs, len(species), arr, (C.c_int * 6)(*bc_B), (C.c_int * 6)(*bc_D)))
        self.time += dt
        self.step += 1

    def _project_gathered(self, fs, species):
        """The divergence projection on a domain that is split between ranks.  The solve is global (a node Laplacian
        over the whole level; the reference hands it to AMReX's MLMG, which communicates inside every V-cycle).  Here
        the blocks of the field and of the species are gathered to EVERY rank, each rank runs the single-block
        projection of the library on the whole level - a second, whole-level context on its own GPU - and keeps its
        own block of the result: the same arithmetic as the single-rank run (bit-identical results), N times the
        work, and one all-gather of the field (10 components) and the species per step over the host's collective.
        Functional, not fast: no BASELINE configuration carries an `elliptic` action."""
        import torch
        import torch.distributed as dist
        backend = dist.get_backend(self.pg)
        dev = "cpu" if backend == "gloo" else self._pack_device()
        if getattr(self, "_proj", None) is None:
            self._proj = MFP(self.p, device=int(torch.cuda.current_device()), rank=0, world=1, stream=self._stream)
        g = self._proj
        lo, n = self.block_lo, self.block_n

        def gather_into(s):
            mine = torch.from_numpy(np.ascontiguousarray(self.download(s))).to(dev)
            parts = [torch.empty_like(mine) for _ in range(self.world)]
            dist.all_gather(parts, mine, group=self.pg)
            full = np.empty((self.p.n_comp(s),) + tuple(self.p.n_cell[::-1]))
            for r, part in enumerate(parts):
                c = rank_coords(r, self.grid)
                o = [c[d] * self.block_n[d] for d in range(3)]
                full[:, o[2]:o[2] + n[2], o[1]:o[1] + n[1], o[0]:o[0] + n[0]] = part.cpu().numpy()
            g.upload(s, full)
        for s in [fs] + list(species):
            gather_into(s)
        arr = (C.c_int * max(1, len(species)))(*species)
        bc_B, bc_D = self.p.elliptic_face_types(fs)
        g._chk(g.L.cmfp_elliptic_project_bc(g.h, fs, len(species), arr, (C.c_int * 6)(*bc_B), (C.c_int * 6)(*bc_D)))
        F = g.download(fs)
        self.upload(fs, F[:, lo[2]:lo[2] + n[2], lo[1]:lo[1] + n[1], lo[0]:lo[0] + n[0]])

    def est_time_step(self):
        """MFP::estTimeStep incl. ParallelDescriptor::ReduceRealMin (MFP_timestep.cpp:4-21)."""
        v = C.c_double()
        self._chk(self.L.cmfp_est_time_step(self.h, C.byref(v)))
        dt = v.value
        if self.world > 1:
            import torch
            import torch.distributed as dist
            t = torch.tensor([dt], dtype=torch.float64, device=self._comm_device())
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.pg)
            dt = float(t.item())
        return dt

    def state_allowed_dt(self, s):
        v = C.c_double()
        self._chk(self.L.cmfp_state_get_allowed_time_step(self.h, s, C.byref(v)))
        return v.value

    def action_allowed_dt(self, a):
        v = C.c_double()
        self._chk(self.L.cmfp_action_get_allowed_time_step(self.h, a, C.byref(v)))
        return v.value

    def run(self, nsteps, dt0=None):
        """computeInitialDt, then advance + computeNewDt (1.1x cap) per step
        (MFP_timestep.cpp:23-119; no stop_time clip)."""
        dts = []
        if self.dt is None:
            self.dt = self.est_time_step() if dt0 is None else dt0
        if self._device_dt and nsteps > 0 and not self._time_bc:
            # the whole loop on the device (cmfp_advance_steps): no host round trip per step
            out, nxt = np.empty(nsteps), C.c_double()
            rc = self.L.cmfp_advance_steps(self.h, self.time, self.dt, nsteps, out.ctypes.data_as(lib.dp), C.byref(nxt))
            if rc == 0:
                for v in out:
                    self.time += float(v)
                self.step += nsteps
                self.dt = nxt.value
                return [float(v) for v in out]
            if rc != 1:
                self._chk(rc)
            self._device_dt = False      # not applicable to this configuration: per-step calls
        for _ in range(nsteps):
            self.advance(self.dt)
            dts.append(self.dt)
            self.dt = self.L.cmfp_compute_new_dt(self.est_time_step(), self.dt)
        return dts

    def last_advance_fused(self):
        return bool(self.L.cmfp_last_advance_fused(self.h))

    def last_advance_path(self):
        """0: staged kernels, 1: tiled fused kernels, 2: gather kernels (1-D / 3-D, or forced with fused=2)."""
        return int(self.L.cmfp_last_advance_fused(self.h))

    def last_source_fusion(self):
        """bit 0: source stage A ran inside the CTU kernels, bit 1: stage C inside the field's (cmfp_last_source_fusion)."""
        return int(self.L.cmfp_last_source_fusion(self.h))

    def last_wrapped_dims(self):
        """bit d: the last fused step wrapped periodic direction d inside its kernels, no ghost fill (cmfp_last_wrapped_dims)."""
        return int(self.L.cmfp_last_wrapped_dims(self.h))

    def last_error_cell(self):
        """(state index, (i, j, k)) of the first offending cell of the last ERR_PHYSICS, or None where the
        failure left no invalid cell-centred value (cmfp_last_error_cell)."""
        import ctypes as C
        st, cell = C.c_int(-1), (C.c_int * 3)()
        if self.L.cmfp_last_error_cell(self.h, C.byref(st), cell) != 0:
            return None
        return st.value, tuple(cell)

    def host_times(self, reset=False):
        """Host seconds spent inside the exchange and dt-reduce hooks (ranks > 1): enqueue cost on the host,
        not device time; everything in them is asynchronous with a device-to-device transport."""
        out = dict(self._host_t)
        if reset:
            for k in self._host_t:
                self._host_t[k] = 0.0
        return out

    def profile_enable(self, on=True):
        self._chk(self.L.cmfp_profile_enable(self.h, int(on)))

    def profile_read(self, slot):
        n, ms = C.c_longlong(), C.c_double()
        self._chk(self.L.cmfp_profile_read(self.h, slot, C.byref(n), C.byref(ms)))
        return n.value, ms.value

    # -- Action API mirror (stage-level access for tests) ------------------
    def update_begin(self):
        self._chk(self.L.cmfp_update_begin(self.h))

    def action_get_data(self, a):
        self._chk(self.L.cmfp_action_get_data(self.h, a, self.time))

    def action_calc_time_derivative(self, a, dt):
        self._chk(self.L.cmfp_action_calc_time_derivative(self.h, a, self.time, dt))

    def action_calc_spatial_derivative(self, a, dt):
        self._chk(self.L.cmfp_action_calc_spatial_derivative(self.h, a, self.time, dt, dt))

    def fill_patch(self, s):
        self._chk(self.L.cmfp_fill_patch(self.h, s))

    # -- device-side transport (cmfp_peer_*) ---------------------------------
    def _peer_setup(self):
        """every rank allocates its arena, the IPC handles are all-gathered (set-up only), every rank maps every
        other rank's arena and names its face neighbours; after a barrier the library moves halos and the dt
        word itself"""
        import torch.distributed as dist
        handle = C.create_string_buffer(abi.IPC_HANDLE_BYTES)
        self._chk(self.L.cmfp_peer_init(self.h, self.rank, self.world, handle, None))
        handles = [None] * self.world
        dist.all_gather_object(handles, bytes(handle.raw), group=self.pg)
        for r in range(self.world):
            if r != self.rank:
                self._chk(self.L.cmfp_peer_open(self.h, r, C.create_string_buffer(handles[r], abi.IPC_HANDLE_BYTES), None))
        for d in range(self.p.dim):
            for side in (0, 1):
                nb = self.neighbour(d, side) if self.grid[d] > 1 else None
                self._chk(self.L.cmfp_peer_set_neighbour(self.h, d, side, -1 if nb is None else nb))
        dist.barrier(group=self.pg)      # every arena is zeroed and mapped before the first message
        self._chk(self.L.cmfp_peer_enable(self.h, 1))

    # -- halo exchange between ranks ----------------------------------------
    def _comm_device(self):
        import torch
        if self._exchange_device is not None:
            return self._exchange_device
        if self.world > 1:
            import torch.distributed as dist
            if dist.get_backend(self.pg) == "gloo":       # the control group of a run on the device-side transport
                return torch.device("cpu")
        return torch.device("cuda", torch.cuda.current_device())

    def neighbour(self, d, side):
        """Rank owning the block next to face (d, side), or None at a physical boundary."""
        c = list(self.coords)
        c[d] += -1 if side == 0 else 1
        if c[d] < 0 or c[d] >= self.grid[d]:
            if not self.p.periodic[d]:
                return None
            c[d] %= self.grid[d]
        return coords_rank(c, self.grid)

    # The three hooks below are all the exchange needs from the device side;
    # tests/test_exchange_gloo.py overrides them with a host emulation to run
    # the protocol on CPU ranks.
    def _halo_count(self, s, d):
        return int(self.L.cmfp_halo_count(self.h, s, d))

    def _halo_pack(self, s, d, side, buf):
        self._chk(self.L.cmfp_halo_pack(self.h, s, d, side, C.c_void_p(buf.data_ptr())))

    def _halo_unpack(self, s, d, side, buf):
        self._chk(self.L.cmfp_halo_unpack(self.h, s, d, side, C.c_void_p(buf.data_ptr())))

    def _buffers(self, s, d, n):
        import torch
        key = (s, d)
        if key not in self._bufs:
            dev = self._pack_device()
            dbuf = [torch.empty(n, dtype=torch.float64, device=dev) for _ in range(4)]
            if self._exchange_device is not None and torch.device(self._exchange_device) != dev:
                # host-staged transport (gloo): host mirrors of the four device buffers (pageable: the copies
                # to and from them are synchronous, which the exchange relies on)
                hbuf = [torch.empty(n, dtype=torch.float64, device=self._exchange_device) for _ in range(4)]
            else:
                hbuf = dbuf
            self._bufs[key] = (dbuf, hbuf)
        return self._bufs[key]

    def _pack_device(self):
        import torch
        return torch.device("cuda", torch.cuda.current_device())

    def _stream_ctx(self):
        """The library enqueues packs, unpacks and cmfp_copy_async on the context's stream; the torch copies
        and collectives of the hooks below are ordered against torch's *current* stream.  With a context on
        a non-default stream that stream is made current for the duration of the hook, so that pack -> send
        and recv -> unpack stay in one stream order ('streams')."""
        import contextlib
        if not getattr(self, "_stream", 0):
            return contextlib.nullcontext()
        import torch
        return torch.cuda.stream(torch.cuda.ExternalStream(self._stream))

    def _dt_reduce(self, user, ctx, ptr):
        """cmfp_dt_reduce_fn: all-reduce(MIN) of the device-resident dt (ReduceRealMin, MFP_timestep.cpp:18),
        enqueued in stream order: no host synchronisation with RCCL."""
        try:
            import time
            import torch
            import torch.distributed as dist
            t0 = time.perf_counter()
            with self._stream_ctx():
                rc = self._dt_reduce_body(torch, dist, ptr)
            self._host_t["dt_reduce"] += time.perf_counter() - t0
            return rc
        except Exception as e:  # never let an exception cross the C boundary
            import traceback
            traceback.print_exc()
            self._exchange_error = e
            return 1

    def _dt_reduce_body(self, torch, dist, ptr):
        if self._dt_buf is None:
            self._dt_buf = torch.empty(1, dtype=torch.float64, device=self._pack_device())
        buf = self._dt_buf
        self._chk(self.L.cmfp_copy_async(self.h, C.c_void_p(buf.data_ptr()), C.c_void_p(ptr), 8))
        if self._exchange_device is not None and torch.device(self._exchange_device) != buf.device:
            h = buf.to(self._exchange_device)     # host-staged transport (gloo)
            dist.all_reduce(h, op=dist.ReduceOp.MIN, group=self.pg)
            buf.copy_(h)
        else:
            dist.all_reduce(buf, op=dist.ReduceOp.MIN, group=self.pg)
        self._chk(self.L.cmfp_copy_async(self.h, C.c_void_p(ptr), C.c_void_p(buf.data_ptr()), 8))
        return 0

    def _exchange(self, user, ctx, s, d):
        """cmfp_exchange_fn: pack both faces, swap with the neighbours, unpack.
        Buffers 0/1 = send lo/hi, 2/3 = recv lo/hi.  s = CMFP_ALL_STATES (-1): the
        faces of every state travel back to back in one message per neighbour."""
        try:
            import time
            import torch.distributed as dist
            t0 = time.perf_counter()
            with self._stream_ctx():
                rc = self._exchange_body(dist, s, d)
            self._host_t["exchange"] += time.perf_counter() - t0
            return rc
        except Exception as e:  # never let an exception cross the C boundary
            import traceback
            traceback.print_exc()
            self._exchange_error = e
            return 1

    def _exchange_body(self, dist, s, d):
        states = list(range(self.n_states)) if s < 0 else [s]
        counts = [self._halo_count(q, d) for q in states]
        offs = [sum(counts[:k]) for k in range(len(counts))]
        dbuf, xbuf = self._buffers(s, d, sum(counts))
        staged = xbuf is not dbuf
        nb_lo, nb_hi = self.neighbour(d, 0), self.neighbour(d, 1)
        for q, off, n in zip(states, offs, counts):
            if nb_lo is not None:
                self._halo_pack(q, d, 0, dbuf[0][off:off + n])
            if nb_hi is not None:
                self._halo_pack(q, d, 1, dbuf[1][off:off + n])
        if staged:
            for k in (0, 1):
                xbuf[k].copy_(dbuf[k])
        ops = []
        if nb_lo is not None and nb_lo == nb_hi:
            # two ranks along a periodic dimension: both messages go to the same peer and are
            # matched in order; the peer's low face (its first send) is my high ghost
            ops = [dist.P2POp(dist.isend, xbuf[0], nb_lo, group=self.pg),
                   dist.P2POp(dist.isend, xbuf[1], nb_hi, group=self.pg),
                   dist.P2POp(dist.irecv, xbuf[3], nb_hi, group=self.pg),
                   dist.P2POp(dist.irecv, xbuf[2], nb_lo, group=self.pg)]
        else:
            if nb_lo is not None:
                ops.append(dist.P2POp(dist.isend, xbuf[0], nb_lo, group=self.pg))
                ops.append(dist.P2POp(dist.irecv, xbuf[2], nb_lo, group=self.pg))
            if nb_hi is not None:
                ops.append(dist.P2POp(dist.isend, xbuf[1], nb_hi, group=self.pg))
                ops.append(dist.P2POp(dist.irecv, xbuf[3], nb_hi, group=self.pg))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        if staged:
            for k in (2, 3):
                dbuf[k].copy_(xbuf[k])
        for q, off, n in zip(states, offs, counts):
            if nb_lo is not None:
                self._halo_unpack(q, d, 0, dbuf[2][off:off + n])
            if nb_hi is not None:
                self._halo_unpack(q, d, 1, dbuf[3][off:off + n])
        return 0
