"""profiles/traffic_<workload>.json (what bench.py's roofline.traffic reads) from the PMC summary of tools/evidence.sh:
python tools/make_traffic.py r04_v1 C2 [C1 ...]  (reads gpurun_out/<tag>/<W>_pmc.txt and <W>_bench.json).
FETCH_SIZE / WRITE_SIZE are in KB per dispatch, from two separate --pmc passes (--kernel-trace only); FETCH_SIZE is left uncorrected
(MI355X_MICROARCH.md: the gfx950 correction depends on the width of the loads, which is mixed here)."""
import json, re, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
tag = sys.argv[1]
for w in sys.argv[2:]:
    txt = (ROOT / "gpurun_out" / tag / f"{w}_pmc.txt").read_text()
    bench = json.loads((ROOT / "gpurun_out" / tag / f"{w}_bench.json").read_text().strip().splitlines()[-1])
    rows = {}
    for line in txt.splitlines():
        m = re.match(r"(\S.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+dispatches\s+(\d+)\s+total\s+([\d.]+)\s+per dispatch\s+([\d.]+)", line)
        if m:
            rows.setdefault(m.group(1).strip(), {})[m.group(2)] = (int(m.group(3)), float(m.group(4)), float(m.group(5)))
    # the solve's kernels: everything but one-off setup kernels; per SOLVE = sum over kernels of total / number of solves of the pass
    setup = ("k_init", "k_pack_persist", "k_dense_init", "rocclr")
    ker = {k: v for k, v in rows.items() if not any(s in k for s in setup) and "FETCH_SIZE" in v and "WRITE_SIZE" in v}
    launches = bench["config"]["kernel_launches_per_solve"]
    main = max(ker, key=lambda k: ker[k]["FETCH_SIZE"][1] + ker[k]["WRITE_SIZE"][1])
    n_solves = ker[main]["FETCH_SIZE"][0] if launches <= 1.5 else None
    if n_solves is None:
        # several launches per solve: solves of the PMC pass = dispatches of the kernel that runs once per Newton iteration / iterations per solve
        n_solves = 25      # bench.py --steps 20 --warmup 5 in tools/evidence.sh
    fetch = sum(v["FETCH_SIZE"][1] for v in ker.values()) / n_solves
    write = sum(v["WRITE_SIZE"][1] for v in ker.values()) / n_solves
    out = {"workload": w, "path": f"device path {bench['config'].get('device_path')} ({tag})", "kernels": sorted(ker),
           "newton_iterations_per_solve": bench["config"]["newton_iter_per_solve"], "kernel_launches_per_solve": launches,
           "fetch_size_kb_raw_per_launch": round(fetch, 2), "write_size_kb_per_launch": round(write, 2),
           "bytes_per_launch": int(round(1024 * (fetch + write))),
           "bytes_per_iteration": int(round(1024 * (fetch + write) / max(bench["config"]["newton_iter_per_solve"], 1e-9))),
           "source": f"profiles/{tag}_{w}_pmc.txt (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, --kernel-trace only); per solve of the workload ({n_solves} solves per pass)",
           "note": "FETCH_SIZE uncorrected (load widths are mixed); on the persistent paths the state of the solve is LDS-resident, so memory-side traffic is far below the algorithmic bytes"}
    (ROOT / "profiles" / f"traffic_{w}.json").write_text(json.dumps(out, indent=1) + "\n")
    print(w, out["bytes_per_launch"], "bytes per solve;", len(ker), "kernels")
