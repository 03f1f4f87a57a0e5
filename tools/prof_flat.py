"""rocprofv3 target: N solves of one flat config (C4 | C5) on whatever path it takes."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from treeqp_amd import capi, problems as P
name = sys.argv[1] if len(sys.argv) > 1 else "C4"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
f = {"C4": P.random_clipping_qp, "C5": P.pruned_chain_qp}[name]()
g = capi.TqGpu(f.nk, f.nx, f.nu).upload(f.as_dict(), f.lambda0)
for _ in range(n):
    r = g.solve(**f.opts)
print(name, "path", g.path, r["status"], r["iter"], r["ls_total"])
g.close()
