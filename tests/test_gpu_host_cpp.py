"""The C++ State/Action mirror (cerberus_amd/host/MFP_host.H) drives the same C
ABI: its demo must reproduce the Python driver's dt sequence and state sums."""
import os
import subprocess

import numpy as np
import pytest

from cerberus_amd import abi, config, problems

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("mode", [[], ["batch"], ["elliptic"], ["mixture"], ["ranks"]],
                         ids=["per_step", "advance_steps", "elliptic_action", "mixture_ions", "two_ranks_peer_transport"])
def test_cpp_host_demo_matches_python_driver(mode):
    from cerberus_amd.mfp import MFP
    exe = os.path.join(ROOT, "cerberus_amd", "host", "host_demo")
    assert os.path.exists(exe), "build it with __graft_entry__.build()"
    out = subprocess.run([exe, "32", "5"] + mode, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    dts = [float(l.split()[1]) for l in out.stdout.splitlines() if l.startswith("dt ")]
    sums = {l.split()[1]: (float(l.split()[3]), float(l.split()[5])) for l in out.stdout.splitlines() if l.startswith("state ")}
    # ["ranks"]: the level cut into two blocks, one context + host thread each, MFP::attach_local_peers (arenas by
    # pointer): the STRICT build is decomposition-independent, so the single-rank run below is the expectation
    cfg = problems.orszag_tang_two_fluid_2d(32)
    if mode == ["elliptic"]:   # the mirror's Elliptic class (apply_change after every step, MFP_advance.cpp:49)
        cfg["actions"]["divergence_cleaning"] = {"type": "elliptic", "projection": 1, "state": "field"}
    if mode == ["mixture"]:    # StateDef::species_mass / species_charge / species_gamma of the mirror
        problems.with_mixture(cfg, "ion", [lambda d: 0.5 + 0.3 * np.sin(d["x"]) * np.cos(d["y"])], [1.0, 3.0], [5 / 3, 1.4], [1.0, 2.0])
    P = config.Problem(cfg, fp_mode=abi.FP_STRICT)
    m = MFP(P)
    m.init_data()
    ref_dts = m.run(5)
    assert dts == ref_dts
    for s, name in enumerate(P.state_names):
        U = m.download(s)
        # the demo accumulates in memory order with a plain loop
        flat = U.ravel()
        tot, sq = 0.0, 0.0
        for v in flat.tolist():
            tot += v
            sq += v * v
        assert sums[name] == (tot, sq), name
    m.close()
