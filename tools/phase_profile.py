"""Diagnostic: per-phase shader cycles of the fused CTU kernel (needs the library
built with mfp_fused.hip compiled -DCMFP_STAMPS for the fast namespace).  Never
quote this build's run time; read the shares."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cerberus_amd import abi, config, problems, lib
from cerberus_amd.mfp import MFP

L = lib.load()
names = ["p0 prim", "sync", "p1 recon+trace", "sync", "p2 sweep1", "sync", "p3 correct", "sync", "p4 sweep2", "sync", "p5 div"]

def run(cfg, label):
    P = config.Problem(cfg, fp_mode=abi.FP_FAST)
    m = MFP(P); m.init_data(); m.run(2)
    out = (C.c_ulonglong * 16)()
    L.cmfp_debug_stamps(out, 1)
    m.run(3)
    m.download(0)
    L.cmfp_debug_stamps(out, 1)
    tot = sum(out[:11])
    print(label)
    for k in range(11):
        print("   %-16s %5.1f %%" % (names[k], 100.0 * out[k] / tot))
    m.close()

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
run(problems.kelvin_helmholtz_2d(n), "hydro minmod HLLC")
cfg = problems.orszag_tang_two_fluid_2d(n)
cfg["actions"] = {"em_fluxes": cfg["actions"]["em_fluxes"]}
run(cfg, "field O6 RH")
