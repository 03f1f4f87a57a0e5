"""Campaign through the JSON wire format (examples/solve_qp_json.cpp): random clipping QPs on random trees written as
qp_in.json (nodes {Q,R,S,q,r,lx,lu,ux,uu}, edges {from,to,A,B,b}, options), solved by the front end `treeqp_solve_json` (one process per
problem, on the GPU), its qp_out.json compared with the CPU oracle on the same problem: verdict, iteration count, x, u, lambda to 1e-10
(the tool prints %.17g).  Usage: python tools/fuzz_json.py [cases] [first seed]"""
import json, subprocess, sys, tempfile, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "oracle")); sys.path.insert(0, str(ROOT / "tests"))
from treeqp_amd import problems as P
import oracle_py as orc
from helpers import flat_to_json

REG = {0: "TREEQP_NO_REGULARIZATION", 1: "TREEQP_ALWAYS_LEVENBERG_MARQUARDT", 2: "TREEQP_ON_THE_FLY_LEVENBERG_MARQUARDT"}


def run(n=50, s0=100):
    exe = ROOT / "treeqp_amd" / "lib" / "treeqp_solve_json"
    stats = {"cases": 0, "fail": 0, "tie": 0}
    t0 = time.perf_counter()
    with tempfile.TemporaryDirectory() as td:
        td = Path(td)
        for c in range(n):
            seed = s0 + c
            rng = np.random.default_rng(seed)
            kind = seed % 3
            if kind == 0:
                f = P.random_shape_qp(seed, depth=int(rng.integers(2, 5)), max_kids=3, nx_range=(1, 6), nu_range=(1, 3), ubound=0.3)
            elif kind == 1:
                f = P.pruned_chain_qp(Nh=int(rng.integers(4, 8)), seed=seed)
            else:
                f = P.random_uniform_tree_qp(seed, nx=int(rng.choice([2, 4, 8])), nu=int(rng.integers(1, 4)), md=int(rng.integers(2, 4)), Nr=(nr := int(rng.integers(2, 4))), Nh=nr + int(rng.integers(0, 3)), ubound=0.4)
            flat = f.as_dict()
            rt = int(rng.integers(1, 3))
            o = dict(maxIter=100, stationarityTolerance=1e-8, lineSearchMaxIter=60, lineSearchBeta=float(rng.choice([0.6, 0.8])), lineSearchGamma=0.1, regType=rt, regTol=1e-6, regValue=1e-8 if rt == 1 else 1e-6)
            wire = dict(solver="tdunes", maxit=o["maxIter"], stationarityTolerance=o["stationarityTolerance"], lineSearchMaxIter=o["lineSearchMaxIter"], lineSearchBeta=o["lineSearchBeta"],
                        lineSearchGamma=o["lineSearchGamma"], checkLastActiveSet=1, clipping=True, regType=REG[rt], regTol=o["regTol"], regValue=o["regValue"])
            (td / "qp_in.json").write_text(json.dumps(flat_to_json(flat, wire)))
            out = subprocess.run([str(exe), str(td / "qp_in.json")], cwd=td, capture_output=True, text=True, timeout=300)
            stats["cases"] += 1
            if out.returncode != 0:
                stats["fail"] += 1
                print(f"FAILED seed {seed}: rc {out.returncode}: {out.stderr[-300:]}", flush=True)
                continue
            d = json.loads(out.stdout)
            ref = orc.solve(flat, orc.default_opts(**o), lambda0=None)
            x = np.concatenate([np.atleast_1d(np.asarray(nd["x"], dtype=float)) for nd in d["solution"]["nodes"]])
            u = np.concatenate([np.atleast_1d(np.asarray(nd["u"], dtype=float)) for nd in d["solution"]["nodes"]] or [np.zeros(0)])
            lam = np.concatenate([np.atleast_1d(np.asarray(e["lam"], dtype=float)) for e in d["solution"]["edges"]])
            err = max(float(np.max(np.abs(a - ref[k]))) / max(1.0, float(np.max(np.abs(ref[k])))) if len(ref[k]) else 0.0 for a, k in ((x, "x"), (u, "u"), (lam, "lam")))
            same = d["info"]["status"] == ref["status"] and d["info"]["num_iter"] == ref["iter"]
            if not (same and err < 1e-10):
                if d["info"]["status"] == 0 and ref["status"] == 0 and err < 1e-5:
                    stats["tie"] += 1
                    print(f"  (rounding-level endgame: seed {seed}: tool {d['info']['num_iter']} oracle {ref['iter']} iterations, difference {err:.1e})", flush=True)
                else:
                    stats["fail"] += 1
                    print(f"MISMATCH seed {seed} [{f.name}]: tool status {d['info']['status']} iterations {d['info']['num_iter']} oracle {ref['status']} / {ref['iter']} err {err:.2e}", flush=True)
            if c % 25 == 24:
                print(f"  {c + 1} cases, {stats['fail']} mismatches, {time.perf_counter() - t0:.0f} s", flush=True)
    print(f"{stats['cases']} problems (seeds {s0}..{s0 + n - 1}) through qp_in.json -> treeqp_solve_json -> qp_out.json against the oracle: {stats['fail']} mismatches; {stats['tie']} rounding-level endgames")
    return stats


if __name__ == "__main__":
    st_ = run(int(sys.argv[1]) if len(sys.argv) > 1 else 50, int(sys.argv[2]) if len(sys.argv) > 2 else 100)
    sys.exit(1 if st_["fail"] else 0)
