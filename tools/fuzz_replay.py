"""Replay one case of a parity campaign (`python tools/fuzz_parity.py N S0`): per-iteration trial counts of the device and of the oracle, the oracle's
error before every iteration.  Usage: python tools/fuzz_replay.py S0 SEED [SEED ...]"""
import os, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "oracle")); sys.path.insert(0, str(ROOT / "tests"))
from treeqp_amd import capi
import oracle_py as orc
from helpers import fuzz_case
s0 = int(sys.argv[1])
for seed in [int(a) for a in sys.argv[2:]]:
    f, opts = fuzz_case(seed, s0)
    ref = orc.solve(f.as_dict(), orc.default_opts(**opts), lambda0=f.lambda0)
    print(f"seed {seed}: {f.name}, opts {opts}")
    print(f"  oracle: status {ref['status']}, {ref['iter']} iterations, {ref['ls_total']} trials")
    print("  oracle trials per iteration:", [int(v) for v in ref["trace_ls"][:ref["iter"]]])
    print("  oracle error before iteration:", [f"{v:.2e}" for v in ref["trace_err"][:ref["iter"] + 1]])
    for path in ("auto", "generic"):
        os.environ["TREEQP_AMD_PATH"] = path
        g = capi.TqGpu(f.nk, f.nx, f.nu).upload(f.as_dict(), f.lambda0)
        os.environ.pop("TREEQP_AMD_PATH")
        r = g.solve(**opts)
        ls = g.iteration_log(256)[0]
        sol = g.solution()
        err = max(float(np.max(np.abs(sol[k] - ref[k]))) for k in ("x", "u", "lam"))
        print(f"  device ({path}, path {g.path}): status {r['status']}, {r['iter']} iterations, {r['ls_total']} trials, last error {r['last_error_norm']:.2e}, max|dev - oracle| {err:.1e}")
        print("    trials per iteration:", [int(v) for v in ls[:r["iter"]]])
        g.close()
    from helpers import ulp_sensitivity
    kd = next((k for k in range(min(r["iter"], ref["iter"])) if int(ls[k]) != int(ref["trace_ls"][k])), min(r["iter"], ref["iter"]))
    print(f"  first difference in iteration {kd}; one-ulp perturbations of the data that change the ORACLE's own trial counts: "
          f"{ulp_sensitivity(orc, f, opts, kd)} of 6 in an iteration <= {kd}, {ulp_sensitivity(orc, f, opts, kd - 1) if kd else 0} of 6 in an iteration < {kd}")
