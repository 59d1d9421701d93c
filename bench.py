#!/usr/bin/env python3
"""Benchmark of the hyperbolic update (BASELINE.json metric: cell-updates/s,
fp64, five-moment two-fluid + Maxwell; % of HBM roofline).

    python bench.py --gpus N --steps K --warmup W

One process per GPU (RANK / LOCAL_RANK / WORLD_SIZE from torch.distributed.run).
A "step" is one full time step of the whole level: Strang source half step,
ghost fill, CTU flux update of every state, source half step, CFL dt estimate
(with the cross-rank min) — on synthetic analytic initial data already resident
in HBM.  Weak scaling: every GPU owns an n x n block.  Rank 0 prints one JSON
line.

The K steps are timed `--reps` times (each block bracketed by a barrier and a
device synchronisation, maximum over ranks); `value` / `ms_per_step` are those of
the median block, all blocks are listed in `rep_ms_per_step`.  At N = 1 the same
run also reports (extra keys): the same workload in STRICT arithmetic, the other
BASELINE configurations that fit one GPU (KH 2048^2, the GEM block 4096 x 2048,
RMI 256^3) with their own roofline entry, and the CPU baseline.
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES = {"ot2f_2d": 320, "ot2f_2d_implicit": 320, "kh_2d": 80, "rmi_3d": 320, "gem_2d": 320}   # 2 x 8 B x Ncomp per cell-update
WORKLOAD_NOTE = {"ot2f_2d": "BASELINE configs[2]: 2D two-fluid Orszag-Tang, plasma5 source",
                 "ot2f_2d_implicit": "BASELINE configs[2], second series: plasma5 solver = 'implicit' (per-cell 9 x 9 solve)",
                 "kh_2d": "BASELINE configs[1]: 2D KH hydro HLLC",
                 "rmi_3d": "BASELINE configs[4]: 3D two-fluid Richtmyer-Meshkov, plasma5 source",
                 "gem_2d": "BASELINE configs[3]: 2D GEM reconnection two-fluid + Maxwell, walls in y; 8 GPUs x 4096x2048 = 8192^2"}
HBM_PEAK_GBS = 8000.0                         # MI355X_MICROARCH.md: 8.0 TB/s spec
PROFILE_ROUND = "r03"                         # profiles/<round>/pmc_<tag>.json: PMC traffic per launch of the kept profiles
PROFILE_TAG = {("ot2f_2d", 4096): "ot2f_4096", ("kh_2d", 2048): "kh_2048", ("gem_2d", 4096): "gem_4096", ("rmi_3d", 256): "rmi3d_256",
               ("rmi_3d", 512): "rmi3d_512"}


def block_cells(workload, n):
    """cells of the per-GPU block: n per side; GEM blocks are n x n/2 (2 x 4 of them make the 8192^2 domain)"""
    if workload == "gem_2d":
        return [n, n // 2]
    return [n] * (3 if workload == "rmi_3d" else 2)


def build_problem(workload, n, grid, fp_mode, corner_transport=True):
    from cerberus_amd import config, problems
    if workload == "ot2f_2d":
        cfg = problems.orszag_tang_two_fluid_2d(n)
    elif workload == "ot2f_2d_implicit":
        cfg = problems.orszag_tang_two_fluid_2d(n, solver="implicit")
    elif workload == "kh_2d":
        cfg = problems.kelvin_helmholtz_2d(n)
    elif workload == "rmi_3d":
        cfg = problems.richtmyer_meshkov_3d(n)
    elif workload == "gem_2d":
        cfg = problems.gem_reconnection_2d(n, n // 2)
    else:
        raise SystemExit(f"unknown workload {workload}")
    if not corner_transport:
        for a in cfg["actions"].values():
            if a["type"] == "CTU":
                a["corner_transport"] = False
    # weak scaling: the domain grows with the process grid, dx and the per-GPU block stay fixed
    # (both initial conditions are periodic with the single-GPU domain length)
    blk = block_cells(workload, n)
    for d in range(cfg["dim"]):
        L = cfg["prob_hi"][d] - cfg["prob_lo"][d]
        cfg["n_cell"][d] = blk[d] * grid[d]
        cfg["prob_hi"][d] = cfg["prob_lo"][d] + L * grid[d]
    return config.Problem(cfg, fp_mode=fp_mode)


def cpu_baseline(workload, seconds=12.0):
    """The CPU restatement (oracle/, kind 'port', g++ -O3 without fast-math) timed on this host's cores on a
    bounded sample of the same workload: 384^2 cells in 32^2 boxes (the reference's amr.max_grid_size in its
    shipped inputs), whole steps until `seconds` of wall time have been spent.  Threads = the cores this job
    may use: OMP_NUM_THREADS where the harness sets it (16 per GPU on the box), else every core of the host."""
    from oracle import oracle as O
    from cerberus_amd import abi
    n = 384
    P = build_problem(workload, n, [1, 1, 1], abi.FP_STRICT)
    visible = os.cpu_count() or 1
    threads = max(1, min(visible, int(os.environ.get("OMP_NUM_THREADS", str(visible)))))
    O.lib().orc_set_threads(threads)
    sim = O.OracleSim(P, max_grid_size=32)
    dt = sim.est_time_step()
    sim.advance(dt)          # untimed first step (page faults, allocations)
    t0 = time.perf_counter()
    steps = 0
    while steps < 200:
        dt = min(sim.est_time_step(), 1.1 * dt)
        sim.advance(dt)
        steps += 1
        if time.perf_counter() - t0 >= seconds:
            break
    el = time.perf_counter() - t0
    out = {"value": n * n * steps / el, "unit": "cell-updates/s", "cores": threads, "kind": "port",
           "host_cpus_visible": visible, "compiler": "g++ -O3 -ffp-contract=off -fopenmp (oracle/Makefile)",
           "sample": f"{workload} {n}x{n}, max_grid_size 32, {steps} steps, {el:.1f} s"}
    # the same port on one core (SURVEY.md §8d asks for both figures): a few steps of a 192^2 sample
    O.lib().orc_set_threads(1)
    n1 = 192
    sim = O.OracleSim(build_problem(workload, n1, [1, 1, 1], abi.FP_STRICT), max_grid_size=32)
    dt = sim.est_time_step()
    sim.advance(dt)
    t0, s1 = time.perf_counter(), 0
    while s1 < 50 and time.perf_counter() - t0 < 5.0:
        dt = min(sim.est_time_step(), 1.1 * dt)
        sim.advance(dt)
        s1 += 1
    out["one_core_value"] = n1 * n1 * s1 / (time.perf_counter() - t0)
    # how well the port itself scales over the cores it was given (round 2: 4.4 x one core on 16 threads, the three
    # FillPatch copies per state and step and the dt reductions were serial; threaded over the boxes since round 3)
    out["parallel_efficiency"] = out["value"] / (threads * out["one_core_value"])
    O.lib().orc_set_threads(threads)
    return out


def pmc_file(workload, n):
    tag = PROFILE_TAG.get((workload, n))
    if not tag:
        return {}, None
    path = os.path.join(ROOT, "profiles", PROFILE_ROUND, f"pmc_{tag}.json")
    if not os.path.exists(path):
        return {}, None
    return json.load(open(path)), os.path.relpath(path, ROOT)


def pmc_of(pmc, *subs):
    """entry of the kept PMC summary whose kernel name contains every substring (largest traffic first)"""
    best = None
    for k, e in pmc.items():
        if all(s in k for s in subs) and (best is None or e.get("hbm_bytes_per_launch", 0) > best.get("hbm_bytes_per_launch", 0)):
            best = e
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--reps", type=int, default=5, help="timed blocks of --steps steps; the median block is reported")
    ap.add_argument("--workload", default="ot2f_2d", choices=sorted(ALGO_BYTES))
    ap.add_argument("--n", "--cells", dest="n", type=int, default=4096, help="cells per side of the per-GPU block")
    ap.add_argument("--fp", default="fast", choices=["fast", "strict"])
    ap.add_argument("--no-fused", action="store_true")
    ap.add_argument("--tiled-hydro", action="store_true", help="diagnostic: LDS-tiled hydro CTU kernel instead of the wave-marching one")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="only the headline workload (profiling runs): no STRICT / other-workload legs, one timed block")
    ap.add_argument("--no-extras-512", action="store_true",
                    help="skip the RMI 512^3 leg (BASELINE configs[4] at full size: 21.5 GB of state, about a minute of host-side set-up)")
    ap.add_argument("--no-sustain", dest="sustain", action="store_false",
                    help="skip the long block (>= 100 steps, >= 0.3 s) that is reported beside the K-step blocks")
    ap.add_argument("--rehearse", action="store_true",
                    help="diagnostic: run the N-rank path on ONE GPU (all ranks on cuda:0, gloo with host-staged halos); not a measurement")
    ap.add_argument("--rehearse-fallback", action="store_true",
                    help="diagnostic, with --rehearse: route halos, dt and barriers through a second gloo group, as the fallback "
                         "from a failed RCCL pre-flight does")
    ap.add_argument("--transport", default="auto", choices=["auto", "peer", "rccl", "gloo"],
                    help="N > 1: halo / dt transport; auto = the library's device-side transport (IPC), else RCCL from the "
                         "callbacks, else host-staged gloo, each after a pre-flight all ranks agree on")
    ap.add_argument("--overlap", type=int, default=1, choices=[0, 1, 2],
                    help="diagnostic: cmfp_set_overlap mode (2 = interior / frame split launches even on an unsplit block)")
    ap.add_argument("--no-corner-transport", action="store_true", help="diagnostic: corner_transport=false (not the headline workload)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from cerberus_amd import abi
    from cerberus_amd.mfp import MFP, block_grid

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hyperbolic update has no CPU fallback")
    if args.rehearse:
        local = 0
    torch.cuda.set_device(local)
    ranks_seen = 1
    kind = "none"             # transport of halos and the dt word: "peer" | "rccl" | "gloo" | "none" (one rank)
    transport = "none (one rank)"
    halo_pg = None            # the group the exchange hooks use ("rccl": the RCCL sub-group; else the default gloo group)
    if world > 1:
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # gloo first, on every rank: it always comes up, and it carries the control traffic of this script
        # (barriers, the maximum over ranks of the timed blocks, agreement on the transport, the IPC handles)
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=600))

        def agree(ok):
            """every rank takes the same branch: the minimum of the ranks' verdicts, over gloo"""
            flag = torch.tensor([1 if ok else 0], dtype=torch.int64)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return int(flag.item()) == 1

        want = args.transport
        if args.rehearse_fallback:
            want = "gloo"
        # 1. the library's device-side transport (cmfp_peer_*): arenas mapped through HIP IPC, no host code per
        #    step.  Pre-flight: three steps of a small two-fluid problem, decomposed like the run itself; any
        #    failure on any rank (IPC refused, a device-side wait that times out) selects the next transport.
        if want in ("auto", "peer"):
            ok = True
            # a partner that never arrives is given up after CMFP_PEER_TIMEOUT seconds per wait: short for the
            # pre-flight (a failed transport then costs seconds, not minutes), the library's default for the run
            old_to = os.environ.get("CMFP_PEER_TIMEOUT")
            os.environ["CMFP_PEER_TIMEOUT"] = "5"
            try:
                P0 = build_problem("ot2f_2d", 64, block_grid(2, world), abi.FP_FAST)
                m0 = MFP(P0, device=local, rank=rank, world=world, stream=0, exchange_device="cpu", transport="peer")
                m0.init_data()
                m0.run(3)
                import numpy as np
                ok = all(bool(np.isfinite(m0.download(s)).all()) for s in range(m0.n_states))
                m0.close()
            except Exception as e:       # noqa: BLE001 - any failure of this transport selects the next one
                print(f"[bench rank {rank}] device-side (IPC) transport pre-flight failed: {e!r}", file=sys.stderr, flush=True)
                ok = False
            finally:
                if old_to is None:
                    os.environ.pop("CMFP_PEER_TIMEOUT", None)
                else:
                    os.environ["CMFP_PEER_TIMEOUT"] = old_to
            if agree(ok):
                kind, transport = "peer", "device-side: IPC-mapped arenas, pack kernels write into the neighbour's slot (cmfp_peer_*)"
                if args.rehearse:
                    transport += " [rehearsal: all ranks on one GPU]"
        # 2. torch.distributed on RCCL with device buffers, from the exchange / dt-reduce callbacks (not with
        #    several ranks on one device: RCCL refuses that).  Pre-flight of exactly the calls the hooks make.
        if kind == "none" and want in ("auto", "rccl") and not args.rehearse:
            ok = True
            try:
                halo_pg = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=600))
                snd = torch.full((1024,), float(rank), dtype=torch.float64, device="cuda")
                rcv = [torch.empty_like(snd), torch.empty_like(snd)]
                lo, hi = (rank - 1) % world, (rank + 1) % world
                if lo == hi:
                    ops = [dist.P2POp(dist.isend, snd, lo, group=halo_pg), dist.P2POp(dist.isend, snd, hi, group=halo_pg),
                           dist.P2POp(dist.irecv, rcv[1], hi, group=halo_pg), dist.P2POp(dist.irecv, rcv[0], lo, group=halo_pg)]
                else:
                    ops = [dist.P2POp(dist.isend, snd, lo, group=halo_pg), dist.P2POp(dist.irecv, rcv[0], lo, group=halo_pg),
                           dist.P2POp(dist.isend, snd, hi, group=halo_pg), dist.P2POp(dist.irecv, rcv[1], hi, group=halo_pg)]
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
                word = torch.tensor([1.0 + rank], dtype=torch.float64, device="cuda")
                dist.all_reduce(word, op=dist.ReduceOp.MIN, group=halo_pg)
                torch.cuda.synchronize()
                if float(rcv[0][0].item()) != float(lo) or float(rcv[1][0].item()) != float(hi) or float(word.item()) != 1.0:
                    raise RuntimeError("RCCL pre-flight: wrong data received")
            except Exception as e:       # noqa: BLE001
                print(f"[bench rank {rank}] RCCL pre-flight failed: {e!r}", file=sys.stderr, flush=True)
                ok = False
            if agree(ok):
                kind, transport = "rccl", "RCCL, device buffers (torch.distributed from the exchange callbacks)"
            else:
                halo_pg = None
        # 3. host-staged buffers over gloo: slow, always available, still a valid measurement
        if kind == "none":
            kind = "gloo"
            transport = ("gloo, host-staged (rehearsal on one GPU)" if args.rehearse
                         else "gloo, host-staged (neither the device-side transport nor RCCL passed its pre-flight)")
        one = torch.ones(1, dtype=torch.int64)
        dist.all_reduce(one, op=dist.ReduceOp.SUM)     # the control group really spans `world` ranks
        ranks_seen = int(one.item())
    reps = 1 if args.no_extras else max(1, args.reps)

    def hbm_copy_gbs():
        """attainable HBM bandwidth of this device: the library's copy probe (16 bytes per lane and access, read +
        write bytes, best of three grid sizes; 1 GiB buffers: far beyond the 256 MB Infinity Cache)"""
        import ctypes as C
        from cerberus_amd import lib
        v = C.c_double()
        rc = lib.load().cmfp_hbm_copy_probe(local, None, 1 << 30, 10, C.byref(v))
        if rc:
            raise SystemExit(f"cmfp_hbm_copy_probe failed: {rc}")
        return v.value

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def measure(workload, n, fp, steps, warmup, nrep, copy_gbs):
        """One workload on this process group: `nrep` timed blocks of `steps` steps after `warmup` steps.
        Returns the JSON fragment (value, ms_per_step, roofline, kernel times)."""
        fp_mode = abi.FP_FAST if fp == "fast" else abi.FP_STRICT
        dim = 3 if workload == "rmi_3d" else 2
        grid = block_grid(dim, world)
        P = build_problem(workload, n, grid, fp_mode, not args.no_corner_transport)
        m = MFP(P, device=local, rank=rank, world=world, stream=0, fused=(3 if args.tiled_hydro else not args.no_fused),
                exchange_device=("cpu" if kind in ("gloo", "peer") else None), overlap=args.overlap, pg=halo_pg,
                transport=("peer" if kind == "peer" else "callback"))
        m.init_data()
        m.run(warmup)
        fused = m.last_advance_path()
        fusion = m.last_source_fusion()   # bit 0: source stage A inside the CTU kernels, bit 1: stage C inside the field's
        sync()
        m.profile_enable(True)
        for s in range(4):
            m.profile_read(s)
        m.host_times(reset=True)
        blocks = []
        for _ in range(nrep):
            sync()
            t0 = time.perf_counter()
            m.run(steps)
            sync()
            el = time.perf_counter() - t0
            t = torch.tensor([el], dtype=torch.float64)
            if world > 1:
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
            blocks.append(float(t.item()))
        prof = [m.profile_read(s) for s in range(4)]
        m.profile_enable(False)
        host = m.host_times()
        # one long block beside the K-step blocks of the contract: at 3 ms per step K = 20 is a 0.06 s region; the
        # sustained figure is one region of >= 100 steps and >= 0.3 s (no per-launch events in it)
        sustained = None
        if args.sustain and nrep > 1:
            n_long = max(steps, 100, int(0.3 / max(1e-6, statistics.median(blocks) / steps)) + 1)
            n_long = min(n_long, 4000)
            sync()
            t0 = time.perf_counter()
            m.run(n_long)
            sync()
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
            if world > 1:
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
            sustained = {"steps": n_long, "seconds": float(t.item()), "ms_per_step": float(t.item()) / n_long * 1e3}
        m.close()
        el = statistics.median(blocks)
        tsteps = steps * nrep
        blk = block_cells(workload, n)
        cells = 1
        for v in blk:
            cells *= v
        res = {"value": cells * world * steps / el, "ms_per_step": el / steps * 1e3,
               "rep_ms_per_step": [b / steps * 1e3 for b in blocks],
               "hbm_roofline_frac_whole_step": ALGO_BYTES[workload] * cells * steps / el / 1e9 / HBM_PEAK_GBS,
               "kernel_ms_per_step": {"ctu_hydro": prof[0][1] / tsteps, "ctu_field": prof[1][1] / tsteps,
                                      "source_rk2": prof[2][1] / tsteps, "ghost_fill": prof[3][1] / tsteps},
               "fused_path": fused, "fp_mode": fp,
               "source_stages": {"A": "prologue of the CTU kernels" if fusion & 1 else "own launch",
                                 "C": "epilogue of the field kernel" if fusion & 2 else "own launch"},
               "workload": f"{workload} {'x'.join(str(v) for v in blk)} per GPU ({WORKLOAD_NOTE[workload]})",
               "process_grid": grid[:dim], "global_cells": list(P.n_cell[:dim]), "time_integration": P.scheme, "cfl": P.cfl}
        if sustained:
            sustained["value"] = cells * world * sustained["steps"] / sustained["seconds"]
            res["sustained"] = sustained
        if world > 1:
            res["host_ms_per_step"] = {k: v / tsteps * 1e3 for k, v in host.items()}
        if rank != 0:
            return res
        # roofline of the dominant kernel family.  Candidates: the fused CTU update of one hydro state (5 comps
        # read + written: 80 B per cell and launch) and the field's update, which in the fused Strang step also
        # applies the second source half step to all three states (20 comps read + written: 320 B per cell; 160 B
        # when the source epilogue is off).  The family with more time per step is `roofline`, the other
        # `roofline_other`.  achieved = algorithmic bytes per launch / mean launch time (HIP events on the
        # library's stream); in 3-D a "launch" of the field family is its plane-tile launches of one step together.
        pmc, pmc_path = pmc_file(workload, n) if fp == "fast" else ({}, None)
        n_hydro = sum(1 for nm in P.state_names if P.cfg["states"][nm]["type"] == "hydro")

        def roof_of(slot, bytes_per_cell, kname, per_update, pmc_subs, states_per_step=1):
            # per_update: launches that together make one update (the 3-D field family); states_per_step: states
            # of the family one step updates (two species can share one marching launch)
            launches, ms = prof[slot]
            if not launches:
                return None
            updates = launches / per_update
            algo = float(bytes_per_cell) * cells * states_per_step * tsteps / updates
            achieved = algo / (ms / updates * 1e-3) / 1e9
            e = None
            traffic = None
            for subs in pmc_subs:
                q = pmc_of(pmc, *subs)
                if q and "hbm_bytes_per_launch" in q:
                    traffic = (traffic or 0.0) + q["hbm_bytes_per_launch"]
                    e = e or q
            r = {"bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": pmc_path,
                 "attainable_copy_GBs": copy_gbs, "frac_of_attainable": achieved / copy_gbs,
                 "algorithmic_bytes_per_launch": algo, "avg_launch_ms": ms / updates, "launches": launches,
                 "ms_per_step": ms / tsteps}
            if e and "valu_busy" in e:
                r["fp64_valu_busy"] = e["valu_busy"]
            return r

        if dim == 3:
            hroof = roof_of(0, 80, "k_ctu3d<HydroF<minmod,HLLC>> (z-marching fused CTU update of one hydro state)", 1, [("k_ctu3d",)], n_hydro)
            per = prof[1][0] / tsteps if prof[1][0] else 1
            epilogue = bool(fusion & 2)
            froof = roof_of(1, 320 if epilogue else 160,
                            "3-D field CTU update, the three marching launches of one step (k_f3m_y1, k_f3m_z, k_f3m_yx)" + (" + Strang source stage C of both species" if epilogue else ""),
                            per, [("k_f3m_y1",), ("k_f3m_z<",), ("k_f3m_yx<",)])
        else:
            hname = ("k_ctu2d<HydroF<minmod,HLLC>> (LDS-tiled fused CTU update of one hydro state)" if args.tiled_hydro
                     else "k_ctu_march<HydroF<minmod,HLLC>> (wave-marching fused CTU update of the hydro states: %d per launch%s)"
                     % (max(1, round(n_hydro * tsteps / max(1, prof[0][0]))),
                        "; Strang source stage A applied to the rows as they are loaded" if fusion & 1 else ""))
            hroof = roof_of(0, 80, hname, 1, [("k_ctu2d<fast::HydroF",) if args.tiled_hydro else ("k_ctu_march",)], n_hydro)
            froof = None
            if prof[1][0] == tsteps:   # one field launch per step (no interior / frame split)
                epilogue = bool(fusion & 2)     # stage C is in the field kernel
                fname = ("k_fld_march<SRC=2> (wave-marching fused CTU update of the vacuum field state + Strang source stage C of both species"
                         + (", stage A applied to the rows as they enter the window)" if fusion & 1 else ")")
                         if epilogue else "k_fld_march (wave-marching fused CTU update of the vacuum field state)")
                froof = roof_of(1, 320 if epilogue else 160, fname, 1, [("k_fld_march",)])
        roof, other = hroof, froof
        if froof and hroof and froof["ms_per_step"] > hroof["ms_per_step"]:
            roof, other = froof, hroof
        res["roofline"], res["roofline_other"] = roof, other
        return res

    copy_gbs = hbm_copy_gbs() if rank == 0 else None
    main_res = measure(args.workload, args.n, args.fp, args.steps, args.warmup, reps, copy_gbs)

    out = None
    if rank == 0:
        out = {
            "metric": "cell-updates/s (fp64, single-fluid hydro)" if args.workload == "kh_2d"
                      else "cell-updates/s (fp64, five-moment 2-fluid+Maxwell)",
            "value": main_res["value"], "unit": "cell-updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": main_res["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic", "reps": reps, "rep_ms_per_step": main_res["rep_ms_per_step"],
            "ranks_seen": ranks_seen, "halo_transport": transport,
            "config": {"workload": main_res["workload"], "process_grid": main_res["process_grid"],
                       "global_cells": main_res["global_cells"], "fp_mode": args.fp,
                       "fused_kernels": bool(main_res["fused_path"]), "corner_transport": not args.no_corner_transport,
                       "time_integration": main_res["time_integration"], "cfl": main_res["cfl"]},
            "hbm_roofline_frac_whole_step": main_res["hbm_roofline_frac_whole_step"],
            "roofline": main_res["roofline"], "roofline_other": main_res["roofline_other"],
            "kernel_ms_per_step": main_res["kernel_ms_per_step"], "source_stages": main_res["source_stages"],
        }
        if "sustained" in main_res:
            out["sustained"] = main_res["sustained"]
        if "host_ms_per_step" in main_res:
            out["host_ms_per_step"] = main_res["host_ms_per_step"]
    # ---- extra legs: single GPU only (they are not part of the scaling contract)
    if world == 1 and not args.no_extras:
        keep = ("value", "ms_per_step", "rep_ms_per_step", "hbm_roofline_frac_whole_step", "kernel_ms_per_step", "roofline",
                "roofline_other", "workload", "fp_mode", "source_stages", "sustained")
        other_fp = "strict" if args.fp == "fast" else "fast"
        r = measure(args.workload, args.n, other_fp, max(5, args.steps // 2), 2, 3, copy_gbs)
        out[other_fp] = {k: r[k] for k in keep if k in r}
        others = {}
        legs = [("kh_2d", 2048, 40), ("gem_2d", 4096, 10), ("rmi_3d", 256, 5), ("ot2f_2d_implicit", 4096, 6)]
        if not args.no_extras_512:
            legs.append(("rmi_3d", 512, 3))
        for w, n, st in legs:
            if (w, n) == (args.workload, args.n):
                continue
            r = measure(w, n, args.fp, st, 2, 3, copy_gbs)
            others[f"{w}:{n}"] = {k: r[k] for k in keep if k in r}
        out["other_workloads"] = others
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.workload)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
