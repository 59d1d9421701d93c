"""Diagnostic: per-component scales and FAST-vs-oracle differences of one drawn option set (tests/test_gpu_random.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from cerberus_amd import abi, config
from cerberus_amd.mfp import MFP
from oracle import oracle as O
from test_gpu_random import draw
seed = int(sys.argv[1])
cfg = draw(seed)
print({k: v for k, v in cfg.items() if k not in ("states", "actions")})
for n, sd in cfg["states"].items():
    print(n, {k: v for k, v in sd.items() if k in ("type", "reconstruction", "flux", "bc", "div_transport", "div_damping")})
print(cfg["actions"])
ref = O.OracleSim(config.Problem(cfg, fp_mode=abi.FP_STRICT)); ref.run(3)
for mode in (abi.FP_FAST, abi.FP_STRICT):
    P = config.Problem(draw(seed), fp_mode=mode)
    m = MFP(P, device=0); m.init_data(); m.run(3)
    print("mode", mode, "path", m.last_advance_path(), "fusion", m.last_source_fusion())
    for s, name in enumerate(P.state_names):
        a, b = m.download(s), ref.get_state(s)
        print(name, "top %.3e" % np.abs(b).max(), " ".join("[%.1e %.1e]" % (np.abs(b[c]).max(), np.abs(a[c] - b[c]).max()) for c in range(b.shape[0])))
    m.close()
