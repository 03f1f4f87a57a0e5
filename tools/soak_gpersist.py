"""Soak of the single-workgroup kernel (g_persist, g_persist_batch) on the GPU box: (a) N batched steps of BASELINE config C5 (256 pruned
trees, one launch per step): every member's verdict, iteration and trial counts equal to the first step's, the solutions of eight members
bit-identical to the first step's at the end; (b) 10 N solves each of three small trees alone: the same.
Usage: python tools/soak_gpersist.py [N = 1000]"""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from treeqp_amd import capi, problems as P

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
t0 = time.perf_counter()
fs = [P.pruned_chain_qp(seed=7 + i) for i in range(256)]
ms = [capi.TqGpu(f.nk, f.nx, f.nu).upload(f.as_dict(), f.lambda0) for f in fs]
opts = dict(fs[0].opts)
first = capi.solve_batch(ms, **opts)
key = [(r["status"], r["iter"], r["ls_total"]) for r in first]
sample = [0, 38, 92, 117, 200, 219, 250, 255]
sol0 = {i: {k: v.copy() for k, v in ms[i].solution().items()} for i in sample}
bad = 0
for step in range(N):
    rs = capi.solve_batch(ms, **opts)
    now = [(r["status"], r["iter"], r["ls_total"]) for r in rs]
    if now != key:
        bad += 1
        if bad < 5:
            print(f"step {step}: verdicts differ at members {[i for i in range(256) if now[i] != key[i]][:8]}", flush=True)
    if step % 250 == 249:
        print(f"  {step + 1} batched steps, {bad} with a differing verdict, {time.perf_counter() - t0:.0f} s", flush=True)
same = all(np.array_equal(ms[i].solution()[k], sol0[i][k]) for i in sample for k in ("x", "u", "lam", "mu_x", "mu_u"))
print(f"C5 batch: {N} steps of 256 trees ({sum(k[1] for k in key)} iterations, {sum(k[2] for k in key)} trials per step; {sum(1 for k in key if k[0] == 0)} members converge): "
      f"{bad} steps with a differing verdict; solutions of members {sample} bit-identical to the first step's: {same}")
for m in ms:
    m.close()
fail = bad > 0 or not same
for label, f in (("thesis example", P.thesis_example()), ("irregular clipping", P.irregular_clipping_qp()), ("random shape seed 5", P.random_shape_qp(5))):
    g = capi.TqGpu(f.nk, f.nx, f.nu).upload(f.as_dict(), f.lambda0)
    o = dict(f.opts) if getattr(f, "opts", None) else {}
    r0 = g.solve(**o)
    s0 = {k: v.copy() for k, v in g.solution().items()}
    badv = 0
    for _ in range(10 * N):
        r = g.solve(**o)
        badv += (r["status"], r["iter"], r["ls_total"]) != (r0["status"], r0["iter"], r0["ls_total"])
    same = all(np.array_equal(g.solution()[k], s0[k]) for k in s0)
    print(f"{label} (path {g.path}): {10 * N} solves, verdict {(r0['status'], r0['iter'], r0['ls_total'])}: {badv} differing verdicts; final solution bit-identical to the first: {same}")
    fail = fail or badv > 0 or not same
    g.close()
print(f"total {time.perf_counter() - t0:.0f} s")
sys.exit(1 if fail else 0)
