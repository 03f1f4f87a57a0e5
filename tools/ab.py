"""A/B timing of experiment builds on the GPU box: python tools/ab.py [--workload C2] [--steps 300] [--trees 1] base wps1 build@VAR=VALUE ...
Each name is a directory under treeqp_amd/lib_var/ (`base` = the product library); one bench.py child per build,
sequentially, two rounds (run-to-run spread)."""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
args = sys.argv[1:]
workload, steps, trees = "C2", "300", "1"
while args and args[0].startswith("--"):
    if args[0] == "--workload": workload = args[1]
    if args[0] == "--steps": steps = args[1]
    if args[0] == "--trees": trees = args[1]
    args = args[2:]
for rnd in range(2):
    for name in args:
        env = dict(os.environ)
        label = name
        if "@" in name:                       # build@VAR=VALUE[,VAR=VALUE]: extra environment for this run
            name, extra = name.split("@", 1)
            for kv in extra.split(","):
                k, v = kv.split("=", 1)
                env[k] = v
        if name != "base":
            env["TREEQP_AMD_LIB"] = str(ROOT / "treeqp_amd" / "lib_var" / name / "libtreeqp_amd.so")
        else:
            env.pop("TREEQP_AMD_LIB", None)
        r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", steps, "--warmup", "30", "--workload", workload,
                            "--no-cpu-baseline", "--no-batched", "--trees", trees], env=env, capture_output=True, text=True, timeout=300)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(f"{label:16s} FAILED rc={r.returncode} {r.stderr[-400:]}", flush=True)
            continue
        d = json.loads(line[-1])
        print(f"{label:28s} {workload} {d['value']:9.0f} it/s  step {1e3 * d['ms_per_step']:7.1f} us  launch {d['roofline']['launch_us']:7.1f} us  "
              f"iters {d['config']['newton_iter_per_solve']:.0f} kkt {d['config']['max_kkt_residual'] if d['config']['max_kkt_residual'] is not None else float('nan'):.1e}", flush=True)
