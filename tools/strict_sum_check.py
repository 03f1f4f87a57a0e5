"""The cases of the parity campaign whose endgame was decided at rounding level (profiles/r03_v2_fuzz_parity_10k.txt), solved again with
reference-order sums (TREEQP_AMD_STRICT_SUM=1): iteration and trial counts of the device against the CPU oracle, with the switch off and on.
Usage: python tools/strict_sum_check.py [profiles/r03_v2_fuzz_parity_10k.txt]"""
import os, re, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "oracle")); sys.path.insert(0, str(ROOT / "tests"))
from treeqp_amd import capi
import oracle_py as orc
from helpers import fuzz_case

src = Path(sys.argv[1]) if len(sys.argv) > 1 else ROOT / "profiles" / "r03_v2_fuzz_parity_10k.txt"
seeds = sorted({int(m.group(1)) for m in re.finditer(r"seed (\d+) path", src.read_text())})
print(f"{len(seeds)} cases from {src.name}")
tot = {"off": 0, "on": 0, "on3": 0}
for seed in seeds:
    f, opts = fuzz_case(seed)
    ref = orc.solve(f.as_dict(), orc.default_opts(**opts), lambda0=f.lambda0)
    row = [f"seed {seed}: oracle {ref['iter']:3d} / {ref['ls_total']:4d}"]
    for label, env in (("off", {}), ("on", {"TREEQP_AMD_STRICT_SUM": "1", "TREEQP_AMD_PATH": "generic"}), ("on3", {"TREEQP_AMD_STRICT_SUM": "1"})):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            g = capi.TqGpu(f.nk, f.nx, f.nu).upload(f.as_dict(), f.lambda0)
        finally:
            for k, v in old.items():
                if v is None: os.environ.pop(k, None)
                else: os.environ[k] = v
        r = g.solve(**opts)
        same = (r["status"], r["iter"], r["ls_total"]) == (ref["status"], ref["iter"], ref["ls_total"])
        tot[label] += same
        row.append(f"{label} (path {g.path}) {r['iter']:3d} / {r['ls_total']:4d} {'==' if same else '!='}")
        g.close()
    print("   ".join(row), flush=True)
print(f"identical verdict, iteration and trial counts: switch off {tot['off']} / {len(seeds)}, on (launch-per-phase kernels) {tot['on']} / {len(seeds)}, on (the path the library picks) {tot['on3']} / {len(seeds)}")
