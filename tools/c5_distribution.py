"""Diagnostic: the 256 trees of BASELINE config C5 one by one -- nodes, iterations, trials, and the time of a solve alone on the
single-workgroup kernel (g_persist, what the batch launch runs) and on the three-launch family (one workgroup per block)."""
import os, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from treeqp_amd import capi, problems as P

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rows = []
for i in range(n):
    f = P.pruned_chain_qp(seed=7 + i)
    t = {}
    for path in ("auto", "generic"):
        os.environ["TREEQP_AMD_PATH"] = path
        g = capi.TqGpu(f.nk, f.nx, f.nu).upload(f.as_dict(), f.lambda0)
        os.environ.pop("TREEQP_AMD_PATH")
        g.event_timing(False)
        for _ in range(3):
            r = g.solve(**f.opts)
        t0 = time.perf_counter()
        for _ in range(5):
            r = g.solve(**f.opts)
        t[path] = ((time.perf_counter() - t0) / 5 * 1e6, g.path)
        g.close()
    rows.append((len(f.nk), r["iter"], r["ls_total"], t["auto"][0], t["generic"][0], t["auto"][1]))
a = np.array(rows, dtype=float)
print("nodes iter trials  us(single workgroup)  us(three launches)  path")
order = np.argsort(-a[:, 3])
for k in order[:12]:
    print(f"{int(a[k,0]):5d} {int(a[k,1]):4d} {int(a[k,2]):6d} {a[k,3]:10.0f} {a[k,4]:10.0f}   {int(a[k,5])}   (tree {k})")
print(f"single workgroup: max {a[:,3].max():.0f} us, mean {a[:,3].mean():.0f}, median {np.median(a[:,3]):.0f}; three launches: max {a[:,4].max():.0f}, mean {a[:,4].mean():.0f}, median {np.median(a[:,4]):.0f}")
print(f"iterations: max {a[:,1].max():.0f}, mean {a[:,1].mean():.1f}; trials: max {a[:,2].max():.0f}, mean {a[:,2].mean():.1f}; sum of single-workgroup times {a[:,3].sum()/1e3:.1f} ms")
