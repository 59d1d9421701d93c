import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import numpy as np
from cerberus_amd import abi, config, problems
from cerberus_amd.mfp import MFP
import torch
def run(pro, hyb):
    os.environ["CMFP_SRC_PROLOGUE"] = "1" if pro else "0"
    cfg = problems.orszag_tang_two_fluid_2d(4096)
    if hyb:
        for n in ("ion", "electron"):
            cfg["states"][n]["flux"] = "HLLE/HLLC"
            cfg["states"][n]["shock_detector"] = {"name": "pressure_jump_detector", "threshold": 0.3}
    P = config.Problem(cfg, fp_mode=abi.FP_FAST)
    m = MFP(P); m.init_data()
    m.run(5)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); m.run(30); torch.cuda.synchronize(); t = time.perf_counter() - t0
    print("hybrid" if hyb else "hllc", "prologue" if pro else "launch", "fusion", m.last_source_fusion(), "ms/step %.3f" % (t / 30 * 1e3), flush=True)
    U = [m.download(s) for s in range(3)]
    m.close()
    return U
a = run(True, True); b = run(False, True)
from helpers import rel_linf
print("prologue vs launch:", [rel_linf(x, y) for x, y in zip(a, b)])
run(True, False)
