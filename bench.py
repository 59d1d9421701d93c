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
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES = {"ot2f_2d": 320, "kh_2d": 80, "rmi_3d": 320, "gem_2d": 320}   # 2 x 8 B x Ncomp per cell-update
WORKLOAD_NOTE = {"ot2f_2d": "BASELINE configs[2]: 2D two-fluid Orszag-Tang, plasma5 source",
                 "kh_2d": "BASELINE configs[1]: 2D KH hydro HLLC",
                 "rmi_3d": "BASELINE configs[4]: 3D two-fluid Richtmyer-Meshkov, plasma5 source",
                 "gem_2d": "BASELINE configs[3]: 2D GEM reconnection two-fluid + Maxwell, walls in y; 8 GPUs x 4096x2048 = 8192^2"}


def block_cells(workload, n):
    """cells of the per-GPU block: n per side; GEM blocks are n x n/2 (2 x 4 of them make the 8192^2 domain)"""
    if workload == "gem_2d":
        return [n, n // 2]
    return [n] * (3 if workload == "rmi_3d" else 2)
HBM_PEAK_GBS = 8000.0                         # MI355X_MICROARCH.md: 8.0 TB/s spec


def build_problem(workload, n, grid, fp_mode, corner_transport=True):
    from cerberus_amd import config, problems
    if workload == "ot2f_2d":
        cfg = problems.orszag_tang_two_fluid_2d(n)
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
    """The CPU restatement (oracle/, kind 'port') timed on this host's cores on a
    bounded sample of the same workload: 384^2 cells in 32^2 boxes (the
    reference's amr.max_grid_size in its shipped inputs), whole steps until
    `seconds` of wall time have been spent."""
    from oracle import oracle as O
    from cerberus_amd import abi
    n = 384
    P = build_problem(workload, n, [1, 1, 1], abi.FP_STRICT)
    threads = max(1, min(16, os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16"))))
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
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="ot2f_2d", choices=sorted(ALGO_BYTES))
    ap.add_argument("--n", "--cells", dest="n", type=int, default=4096, help="cells per side of the per-GPU block")
    ap.add_argument("--fp", default="fast", choices=["fast", "strict"])
    ap.add_argument("--no-fused", action="store_true")
    ap.add_argument("--tiled-hydro", action="store_true", help="diagnostic: LDS-tiled hydro CTU kernel instead of the wave-marching one")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="only the headline workload (profiling runs): no STRICT / other-workload legs")
    ap.add_argument("--rehearse", action="store_true",
                    help="diagnostic: run the N-rank path on ONE GPU (all ranks on cuda:0, gloo with host-staged halos); not a measurement")
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
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))

    def hbm_copy_gbs():
        """attainable HBM bandwidth of this device: a 1 GiB device-to-device copy (read + write bytes)"""
        a = torch.empty(1 << 27, dtype=torch.float64, device="cuda")
        b_ = torch.empty_like(a)
        b_.copy_(a)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            b_.copy_(a)
        e1.record()
        torch.cuda.synchronize()
        return 10 * 2 * a.numel() * 8 / (e0.elapsed_time(e1) * 1e-3) / 1e9

    fp_mode = abi.FP_FAST if args.fp == "fast" else abi.FP_STRICT
    dim = 3 if args.workload == "rmi_3d" else 2
    grid = block_grid(dim, world)
    P = build_problem(args.workload, args.n, grid, fp_mode, not args.no_corner_transport)
    m = MFP(P, device=local, rank=rank, world=world, stream=0, fused=(3 if args.tiled_hydro else not args.no_fused),
            exchange_device=("cpu" if args.rehearse and world > 1 else None), overlap=args.overlap)
    m.init_data()

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    m.run(args.warmup)
    fused = m.last_advance_fused()
    sync()
    m.profile_enable(True)
    for s in range(4):
        m.profile_read(s)
    t0 = time.perf_counter()
    m.run(args.steps)
    sync()
    el = time.perf_counter() - t0
    prof = [m.profile_read(s) for s in range(4)]
    m.profile_enable(False)

    t = torch.tensor([el], dtype=torch.float64, device="cpu" if args.rehearse else "cuda")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    el = float(t.item())

    blk = block_cells(args.workload, args.n)
    cells_per_gpu = 1
    for v in blk:
        cells_per_gpu *= v
    total_cells = cells_per_gpu * world
    value = total_cells * args.steps / el

    out = None
    if rank == 0:
        # roofline of the dominant kernel.  Candidates: the fused CTU update of one hydro state (5 comps
        # read + written: 80 B per cell) and, in the 2-D two-fluid workloads, the field's CTU launch, which
        # also applies the second source half step to all three states (20 comps read + written: 320 B per
        # cell; 160 B when the source epilogue is off).  The one with more time per step is reported as
        # `roofline`, the other as `roofline_other`.
        tpath = os.path.join(ROOT, "profiles", "r01_traffic.json")
        tj = json.load(open(tpath)).get(f"{args.workload}:{args.n}", {}) if os.path.exists(tpath) else {}
        copy_gbs = hbm_copy_gbs()

        def roof_of(slot, bytes_per_cell, kname, traffic_key):
            launches, ms = prof[slot]
            if not launches:
                return None
            algo = float(bytes_per_cell) * cells_per_gpu
            achieved = algo / (ms / launches * 1e-3) / 1e9
            return {"bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": tj.get(traffic_key), "attainable_copy_GBs": copy_gbs,
                    "frac_of_attainable": achieved / copy_gbs, "algorithmic_bytes_per_launch": algo,
                    "avg_launch_ms": ms / launches, "launches": launches, "ms_per_step": ms / args.steps}

        if dim == 3:
            hname = "k_ctu3d<HydroF<minmod,HLLC>> (z-marching fused CTU update of one hydro state)"
        elif args.tiled_hydro:
            hname = "k_ctu2d<HydroF<minmod,HLLC>> (LDS-tiled fused CTU update of one hydro state)"
        else:
            hname = "k_ctu_march<HydroF<minmod,HLLC>> (wave-marching fused CTU update of one hydro state)"
        roof = roof_of(0, 80, hname, "ctu_hydro_bytes_per_launch")
        roof_other = None
        if dim == 2 and prof[1][0] == args.steps:   # one field launch per step (no interior / frame split)
            epilogue = prof[2][0] == args.steps     # one separate source launch per step: stage C is in the field kernel
            fname = ("k_ctu2d<FieldF<O6>,SRC=2> (LDS-tiled fused CTU update of the field state + Strang source stage C of both species)"
                     if epilogue else "k_ctu2d<FieldF<O6>> (LDS-tiled fused CTU update of the field state)")
            froof = roof_of(1, 320 if epilogue else 160, fname, "ctu_field_bytes_per_launch")
            if froof and roof and froof["ms_per_step"] > roof["ms_per_step"]:
                roof, roof_other = froof, roof
            else:
                roof_other = froof
        step_bytes = ALGO_BYTES[args.workload] * cells_per_gpu
        out = {
            "metric": "cell-updates/s (fp64, single-fluid hydro)" if args.workload == "kh_2d"
                      else "cell-updates/s (fp64, five-moment 2-fluid+Maxwell)",
            "value": value, "unit": "cell-updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": el / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.workload} {'x'.join(str(v) for v in blk)} per GPU ({WORKLOAD_NOTE[args.workload]})",
                       "process_grid": grid[:dim], "global_cells": list(P.n_cell[:dim]), "fp_mode": args.fp,
                       "fused_kernels": bool(fused), "corner_transport": not args.no_corner_transport, "time_integration": P.scheme, "cfl": P.cfl},
            "hbm_roofline_frac_whole_step": (step_bytes * args.steps / (el)) / 1e9 / HBM_PEAK_GBS,
            "roofline": roof, "roofline_other": roof_other,
            "kernel_ms_per_step": {"ctu_hydro": prof[0][1] / args.steps, "ctu_field": prof[1][1] / args.steps,
                                   "source_rk2": prof[2][1] / args.steps, "ghost_fill": prof[3][1] / args.steps},
        }
    m.close()
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
