"""Diagnostic: after k Newton iterations (maxIter = k), is the device's iterate bit-identical to the CPU oracle's?  (TREEQP_AMD_STRICT_SUM=1)
Usage: python tools/strict_bits.py SEED [SEED ...]"""
import os, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "oracle")); sys.path.insert(0, str(ROOT / "tests"))
from treeqp_amd import capi
import oracle_py as orc
from helpers import fuzz_case
os.environ["TREEQP_AMD_STRICT_SUM"] = "1"
os.environ["TREEQP_AMD_PATH"] = "generic"
for seed in [int(a) for a in sys.argv[1:]]:
    f, opts = fuzz_case(seed)
    print(f"seed {seed}: {len(f.nk)} nodes, opts {opts}")
    for k in range(0, 6):
        o = dict(opts); o["maxIter"] = max(k, 1)
        if k == 0: o["maxIter"] = 1
        ref = orc.solve(f.as_dict(), orc.default_opts(**o), lambda0=f.lambda0)
        g = capi.TqGpu(f.nk, f.nx, f.nu).upload(f.as_dict(), f.lambda0)
        r = g.solve(**o)
        sol = g.solution()
        row = [f"  maxIter {o['maxIter']}: device {r['status']} {r['iter']}/{r['ls_total']} oracle {ref['status']} {ref['iter']}/{ref['ls_total']}"]
        for key in ("x", "u", "lam", "mu_x"):
            a, b = np.asarray(sol[key]), np.asarray(ref[key])
            same = int((a.view(np.uint64) == b.view(np.uint64)).sum()) if a.size else 0
            row.append(f"{key}: {same}/{a.size} bits equal, max diff {np.max(np.abs(a - b)) if a.size else 0:.2e}")
        row.append(f"fval dev {r['last_fval']!r} orc {ref.get('fval', ref.get('last_fval'))!r}")
        print("  ".join(row), flush=True)
        g.close()
