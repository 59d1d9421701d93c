#!/usr/bin/env python3
"""Writes tests/golden/oracle_nstep.json: for the reduced-size BASELINE configurations of
tests/helpers.py, the dt sequence (hex floats) and a SHA-256 of every state's conserved data after N
steps of the CPU restatement (oracle/, STRICT arithmetic).  The fixture pins the orchestration, ghost
fill and integrator order of the oracle against silent change, and gives the GPU's STRICT path a
committed vector to match without running the oracle.  Regenerate only on a deliberate change of the
algorithm:   python tools/make_golden_nstep.py"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.float64).tobytes()).hexdigest()


def main():
    from cerberus_amd import abi, config
    from helpers import small_cases, survey_cases
    from oracle import oracle as O
    out = {"_comment": "written by tools/make_golden_nstep.py from the CPU restatement (STRICT); sha256 of float64 bytes"}
    for name, cfg, steps in small_cases() + survey_cases():
        P = config.Problem(cfg, fp_mode=abi.FP_STRICT)
        sim = O.OracleSim(P, max_grid_size=32 if "@" in name else 0)   # results do not depend on the box chopping (tested)
        dts = sim.run(steps)
        out[name] = {"steps": steps, "dts": [float(v).hex() for v in dts],
                     "states": {P.state_names[s]: {"sha256": digest(sim.get_state(s)), "shape": list(sim.get_state(s).shape),
                                                   "sum": float(np.sum(sim.get_state(s))).hex()}
                                for s in range(len(P.state_names))}}
    path = os.path.join(ROOT, "tests", "golden", "oracle_nstep.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
