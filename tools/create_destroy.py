"""Diagnostic: create / solve / destroy many mirrors (handles, pinned buffers, events and streams must all be released)."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch
from treeqp_amd import capi, problems as P
from helpers import product_qp_from_lti
free0 = torch.cuda.mem_get_info()[0]
probs = [P.linear_chain(2, 9, 9), P.spring_mass(), None, None]
for i in range(300):
    j = i % 4
    if j < 2:
        p = probs[j]; qp = product_qp_from_lti(capi, p); flat = qp.flat(); l0 = p.lambda0
    elif j == 2:
        f = P.pruned_chain_qp(seed=7 + i); flat = f.as_dict(); l0 = None
    else:
        f = P.random_clipping_qp(nx=10, nu=4, md=3, levels=4); flat = f.as_dict(); l0 = None
    g = capi.TqGpu(flat["nk"], flat["nx"], flat["nu"]).upload(flat, l0)
    r = g.solve(maxIter=10) if j == 3 else g.solve()
    assert r["status"] in (0, 1), (i, r)
    g.close()
    if i in (49, 99, 199):
        print(f"after {i+1} cycles: free {torch.cuda.mem_get_info()[0]/2**20:.0f} MiB", flush=True)
free1 = torch.cuda.mem_get_info()[0]
print(f"300 create/solve/destroy cycles ok; device memory free before {free0/2**20:.0f} MiB, after {free1/2**20:.0f} MiB")
