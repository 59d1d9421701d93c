import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import numpy as np
from cerberus_amd import abi, config, problems
from cerberus_amd.mfp import MFP
from oracle import oracle as O
cfg = problems.brio_wu_two_fluid_1d(2048)
P = config.Problem(cfg, fp_mode=abi.FP_STRICT)
ref = O.OracleSim(P, max_grid_size=32)
for fused in (True, False):
    ref = O.OracleSim(P, max_grid_size=32)
    m = MFP(P, fused=fused); m.init_data()
    dt = None
    for step in range(100):
        dt = ref.est_time_step() if dt is None else min(ref.est_time_step(), 1.1*dt)
        ref.advance(dt); m.advance(dt)
        bad = False
        for s in range(3):
            a, b = m.download(s), ref.get_state(s)
            if not np.array_equal(a, b):
                idx = np.argwhere(a != b)
                print("fused", fused, "step", step, "state", P.state_names[s], "ndiff", len(idx), "first", idx[0], float(a[tuple(idx[0])]).hex(), float(b[tuple(idx[0])]).hex())
                c, _, _, i = idx[0]
                print(" neighbours gpu", [float(v).hex() for v in a[c,0,0,i-3:i+4]])
                print(" neighbours ref", [float(v).hex() for v in b[c,0,0,i-3:i+4]])
                bad = True
        if bad: break
    m.close()
