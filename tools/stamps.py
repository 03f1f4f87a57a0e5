"""Diagnostic: in-kernel phase stamps of the fused kernels (workgroup 0) on the GPU box.
Needs a library built with the stamps compiled in: TQ_DEFS=-DTQ_STAMPS python treeqp_amd/build.py --force
Usage: TREEQP_AMD_STAMPS=2 python tools/stamps.py [Nr]"""
import ctypes as C
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("TREEQP_AMD_STAMPS", "2")      # persistent path: the launch-relative iteration to stamp
from treeqp_amd import capi, problems as P

Nr = int(sys.argv[1]) if len(sys.argv) > 1 else 9
p = P.spring_mass() if os.environ.get("STAMPS_CASE") == "C1" else P.linear_chain(2, Nr, Nr)
nk = p.nk()
nx = np.full(p.Nn, p.nx, dtype=np.int32)
nu = np.where(nk > 0, p.nu, 0).astype(np.int32)
qp = capi.TreeQp(nx, nu, nk).fill_lti(p)
g = capi.TqGpu(nk, nx, nu).upload(qp.flat(), p.lambda0)
kw = {}
if os.environ.get("STAMPS_TOL"):
    kw["stationarityTolerance"] = float(os.environ["STAMPS_TOL"])
for _ in range(5):
    r = g.solve(maxIter=1) if g.path != 2 else g.solve(**kw)   # tiered: one real iteration; persistent: full solve, iteration $TREEQP_AMD_STAMPS
print(r)
buf = np.zeros(8 * 32 * 2, dtype=np.uint64)
capi.lib().tqgpu_get_stamps(g.h, buf.ctypes.data_as(C.POINTER(C.c_ulonglong)), len(buf))
st = buf.reshape(8, 32, 2).astype(np.int64)
t_first = min(int(st[k][0, 1]) for k in range(8) if st[k][0, 1] > 0)
LAUNCH0 = min([int(st[k][25, 1]) for k in range(8) if st[k][25, 1] > 0] or [0])
for kern in range(8):
    name = f"kernel#{kern} (back tiers, top, fwd tiers, stage in launch order)"
    s = st[kern]
    n = int((s[:, 1] > 0).sum())
    if n < 2:
        continue
    if kern == 7:
        print('fine stamps (cycles) load_rows / sub_children / factor / store_factor / schur:', [int(v) for v in np.diff(s[:6, 0])])
        continue
    if s[25, 1] > 0:
        names = ["start", "state loaded", "first sweep done", "left the loop", "verdict sent to host", "state written back"]
        w = s[25:31, 1].astype(np.int64)
        print(f"   launch-level stamps of workgroup {kern} (100 MHz clock, us after the earliest start):", ", ".join(f"{names[i]} {(w[i] - LAUNCH0) * 0.01:.2f}" for i in range(6) if w[i] > 0))
        s = s.copy(); s[25:31] = 0
    if s[20, 0] > 0 and s[24, 0] > 0:
        print("   fine stamps of level t = 1, wave 0 (cycles, each includes one stamp): assemble / factor / store_factor / schur:", [int(v) for v in np.diff(s[20:25, 0])])
        s = s.copy(); s[20:25] = 0
        n = int((s[:20, 1] > 0).sum())
    if s[31, 0] > 0:
        print(f"   (cost of one stamp: {int(s[31, 0]) - int(s[0, 0])} cycles)")
        s = s.copy(); s[31] = 0
        n = int((s[:31, 1] > 0).sum())
    cyc = np.diff(s[:n, 0])
    wall = np.diff(s[:n, 1]) * 10.0          # 100 MHz -> ns
    print(name, "phases:", n - 1, f" starts at +{(int(s[0, 1]) - t_first) * 0.01:.2f} us")
    for i in range(n - 1):
        print(f"   {i:2d}: {wall[i] / 1e3:8.2f} us  {cyc[i]:8d} cycles  ({cyc[i] / max(wall[i], 1):.2f} GHz)")
    print(f"   total {wall.sum() / 1e3:.2f} us")
# placement census of the persistent launch (TQ_STAMPS builds): workgroup -> (XCC, SE, CU)
big = np.zeros(8 * 32 * 2 + 1024, dtype=np.uint64)
capi.lib().tqgpu_get_stamps(g.h, big.ctypes.data_as(C.POINTER(C.c_ulonglong)), len(big))
cen = big[8 * 32 * 2:]
place = {}
for wg_id, v in enumerate(cen):
    v = int(v)
    if v == 0:
        continue
    hw, xcc = v & 0xFFFFFFFF, (v >> 32) & 0xF
    cu, sh, se = (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 0x7
    place.setdefault((xcc, se, sh, cu), []).append(wg_id)
shared = {k: v for k, v in place.items() if len(v) > 1}
print(f"placement: {sum(len(v) for v in place.values())} workgroups on {len(place)} distinct (xcc, se, sh, cu); shared CUs: {len(shared)}")
for k, v in sorted(shared.items())[:40]:
    print("   CU", k, "holds workgroups", v)
