"""Writes tests/golden/thermal_rmi_tree.json: the reference's shipped input Exec/testing/Thermal-RMI (TRMI.inputs +
problem_definition.lua) AS LOADED by cerberus_amd/lua_input.py, at the level-0 resolution the inputs file gives
(amr.n_cell = 64 8), as a key / value tree: ParmParse geometry, the Lua globals the path consumes, the option
tables of the on-path states and actions, and every `value` entry evaluated at the cell centres (numbers stay
numbers, Lua functions become arrays [k][j][i]).  No Lua text is stored.  Run in the container (the reference tree
does not travel to the GPU box):

    python tools/make_trmi_fixture.py

tests/test_trmi_fixture.py rebuilds the problem from the tree and runs it on the HIP path against the oracle, and -
where the reference tree is present - checks the tree against a fresh load."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from cerberus_amd import config, lua_input  # noqa: E402

SRC = "/root/reference/Exec/testing/Thermal-RMI/TRMI.inputs"
OUT = os.path.join(ROOT, "tests", "golden", "thermal_rmi_tree.json")


def tree_of(path=SRC):
    cfg, info = lua_input.load_inputs(path)
    P = config.Problem(cfg)
    dat = P.cell_centres()
    shape = dat["x"].shape

    def plain(v):
        if callable(v):
            return {"__array__": np.broadcast_to(np.asarray(v(dat), dtype=np.float64), shape).tolist()}
        if isinstance(v, dict):
            return {k: plain(x) for k, x in v.items()}
        if isinstance(v, (list, tuple)):
            return [plain(x) for x in v]
        if isinstance(v, (np.floating, np.integer)):
            return v.item()
        return v
    tree = {k: plain(v) for k, v in cfg.items() if k not in ("states", "actions", "_ignored")}
    tree["states"] = {n: plain(sd) for n, sd in cfg["states"].items()}
    tree["actions"] = {n: plain(ad) for n, ad in cfg["actions"].items()}
    tree["ignored"] = list(info["ignored"])
    tree["source"] = "Exec/testing/Thermal-RMI/TRMI.inputs (amr.n_cell as shipped), read by cerberus_amd/lua_input.py"
    return tree


def problem_cfg(tree):
    """the dict config.Problem takes, rebuilt from the tree (arrays become functions of the cell-centre dict that
    return the part of the whole-level array at the centres asked for)"""
    dim = int(tree["dim"])
    lo = [float(v) for v in tree["prob_lo"]]
    dx = [(float(tree["prob_hi"][d]) - lo[d]) / int(tree["n_cell"][d]) for d in range(dim)]

    def block_of(arr, d):
        """the part of the whole-level array at the cell centres asked for (a rank's block: located by its first centre)"""
        shape = d["x"].shape
        sl = [slice(None)] * 3
        for ax, key in ((2, "x"), (1, "y"), (0, "z")):
            e = 2 - ax
            if e >= dim:
                continue
            first = float(np.asarray(d[key]).reshape(-1)[0])
            i0 = int(round((first - lo[e]) / dx[e] - 0.5))
            sl[ax] = slice(i0, i0 + shape[ax])
        return arr[tuple(sl)]

    def live(v):
        if isinstance(v, dict) and "__array__" in v:
            arr = np.asarray(v["__array__"], dtype=np.float64)
            return lambda d, arr=arr: block_of(arr, d)
        if isinstance(v, dict):
            return {k: live(x) for k, x in v.items()}
        if isinstance(v, list):
            return [live(x) for x in v]
        return v
    cfg = {k: live(v) for k, v in tree.items() if k not in ("ignored", "source")}
    return cfg


if __name__ == "__main__":
    t = tree_of()
    with open(OUT, "w") as f:
        json.dump(t, f)
    print("wrote", OUT, os.path.getsize(OUT), "bytes; states", sorted(t["states"]), "actions", sorted(t["actions"]))
