#!/usr/bin/env python3
"""FAST arithmetic against the CPU oracle as a function of the step count (run on the GPU box):
relative L-infinity of every state of BASELINE configs 1 and 3 (Brio-Wu 2048 cells; two-fluid Orszag-Tang
256^2) after 1, 2, 5, 10, 20, 50 and 100 steps.  The north-star bound is 1e-12 "after N steps"; this
records how much of it FAST (seeded reciprocals / roots, FMA contraction) uses and whether it grows.
Writes profiles/<round>/fast_drift.json.   usage: python tools/fast_drift.py profiles/r02"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(outdir):
    from cerberus_amd import abi, config, problems
    from cerberus_amd.mfp import MFP
    from helpers import rel_linf
    from oracle import oracle as O
    marks = [1, 2, 5, 10, 20, 50, 100]
    out = {"_comment": "rel L-inf (max|a-b| / max|b| per state) of the FAST HIP path against the STRICT CPU oracle; "
                       "STRICT HIP is bit-identical to the oracle at every mark (checked here too)"}
    for name, cfg in (("brio_wu_1d@2048", problems.brio_wu_two_fluid_1d(2048)), ("ot2f_2d@256", problems.orszag_tang_two_fluid_2d(256))):
        P = config.Problem(cfg, fp_mode=abi.FP_STRICT)
        ref = O.OracleSim(P, max_grid_size=32)
        f = MFP(config.Problem(cfg, fp_mode=abi.FP_FAST))
        s_ = MFP(P)
        f.init_data()
        s_.init_data()
        dt, done, rows = None, 0, []
        for mk in marks:
            for _ in range(mk - done):
                dt = ref.est_time_step() if dt is None else min(ref.est_time_step(), 1.1 * dt)
                ref.advance(dt)
            f.run(mk - done)
            s_.run(mk - done)
            done = mk
            errs = {P.state_names[k]: float(rel_linf(f.download(k), ref.get_state(k))) for k in range(f.n_states)}
            strict_ok = all(np.array_equal(s_.download(k), ref.get_state(k)) for k in range(f.n_states))
            rows.append({"steps": mk, "fast_rel_linf": errs, "strict_bit_identical": bool(strict_ok)})
            print(name, mk, errs, strict_ok, flush=True)
        out[name] = rows
        f.close()
        s_.close()
    os.makedirs(outdir, exist_ok=True)
    json.dump(out, open(os.path.join(outdir, "fast_drift.json"), "w"), indent=1)
    worst = max(v for k, rows in out.items() if k != "_comment" for r in rows for v in r["fast_rel_linf"].values())
    print("worst", worst)
    return 0 if worst <= 1e-12 else 1


if __name__ == "__main__":
    sys.exit(main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out")))
