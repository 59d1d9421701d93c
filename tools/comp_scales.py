"""Diagnostic: per conserved component, the largest reference value and the largest FAST - STRICT(staged) difference
after a few steps (what a per-component relative L-inf is made of)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cerberus_amd import abi, config, problems
from cerberus_amd.mfp import MFP

def run(cfg, mode, fused, steps):
    P = config.Problem(cfg, fp_mode=mode)
    m = MFP(P, device=0, fused=fused)
    m.init_data()
    m.run(steps)
    out = [m.download(s) for s in range(m.n_states)]
    m.close()
    return P, out

res = {}
for name, make, steps in [("ot256x20", lambda: problems.orszag_tang_two_fluid_2d(256), 20),
                          ("ot1024x4", lambda: problems.orszag_tang_two_fluid_2d(1024), 4),
                          ("ot4096x4", lambda: problems.orszag_tang_two_fluid_2d(4096), 4),
                          ("gem512x256x10", lambda: problems.gem_reconnection_2d(512, 256), 10),
                          ("rmi64x10", lambda: problems.richtmyer_meshkov_3d(64), 10)]:
    P, a = run(make(), abi.FP_FAST, True, steps)
    _, b = run(make(), abi.FP_STRICT, False, steps)
    for s, nm in enumerate(P.state_names):
        top = np.abs(b[s]).max()
        rows = []
        for c in range(b[s].shape[0]):
            rows.append((float(np.abs(b[s][c]).max()), float(np.abs(a[s][c] - b[s][c]).max())))
        res[f"{name}/{nm}"] = {"top": float(top), "comp_max_and_abs_err": rows}
        print(name, nm, "top %.3e" % top, " ".join("[%.1e %.1e]" % r for r in rows), flush=True)
json.dump(res, open(sys.argv[1], "w"), indent=1)
