#!/usr/bin/env python3
"""bench.py -- dual-Newton iterations/s of the tdunes hot path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line on rank 0.
For N > 1 it is launched under torch.distributed.run (one rank per GPU, RCCL); started WITHOUT the
launcher (`python bench.py --gpus N`, WORLD_SIZE unset) it starts the launcher itself as a child
process -- before anything touches a GPU -- and forwards the child's JSON line.

* step      = one full tdunes solve of the workload from the same resident lambda0 (the reference's
              own timing protocol: examples/spring_mass_dual_newton_tree.c:135-140); QP data,
              index tables and lambda0 are resident in HBM before the timed region starts.
* value     = Newton iterations of ALL ranks / wall time of the K steps (max over ranks).
* workload  = BASELINE.json configs[1] ("C2", the configuration the metric is quoted on): linear-chain
              spring-mass tree nx=8, nu=3, 10 levels, branching 2 -> 1023 nodes.
              --workload C1|C3|C4|C5 run the other BASELINE configurations through the same harness
              (C5: a batch of pruned scenario trees per step, the shape of fault_tolerance.c:486-530).
* N > 1     = default `--mode batch`: one tree per GPU (independent scenario trees, weak scaling, no
              data-path collective), and -- in the same line, under "sharded" -- ONE C3 tree partitioned by
              subtrees over the N ranks with two small RCCL all-gathers per Newton iteration (SURVEY.md
              §8e; strong scaling, latency-bound by construction).  `--mode shard` makes that the headline.
* roofline  = algorithmic bytes of the Newton iterations (closed form of SURVEY.md §8d, evaluated
              by tqgpu_iteration_cost) / device time between HIP events recorded on the solver's own
              stream around each solve (tqgpu_get_device_times), against the 8 TB/s HBM3E peak.  The event
              pairs are recorded in a second region of the same solves right after the timed one: the timed
              region enqueues a solve as the bare kernel launch it is (an event pair is two more queue packets).
* critical_path = the dependent chain that actually bounds a solve: tree levels x measured floor of a level step in
              isolation (tools/microbench/level_bench, profiles/r03_v1_level_bench.txt) + tier hand-overs, against the
              measured period of a pass (solves stopped after 1 and 2 iterations).
* cpu_baseline = the CPU oracle ("port": restatement of the reference algorithm, NOT BLASFEO
              HIGH_PERFORMANCE) on the same workload, min over repetitions, rank 0 at N=1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
for p in (ROOT, ROOT / "oracle"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
CLOCK_GHZ = 2.4              # shader clock the level floors were measured at (cycles -> us)


def make_workload(name: str):
    """-> (list of flat QPs solved per step, description, solver options)"""
    from treeqp_amd import capi, problems as P

    def lti(p):
        nk = p.nk()
        nx = np.full(p.Nn, p.nx, dtype=np.int32)
        nu = np.where(nk > 0, p.nu, 0).astype(np.int32)
        qp = capi.TreeQp(nx, nu, nk).fill_lti(p)      # QP built through the reference-compatible host API
        return dict(flat=qp.flat(), lambda0=p.lambda0, qp=qp, nodes=int(p.Nn))

    if name == "C2":
        return [lti(P.linear_chain(2, 9, 9))], "linear_chain nx=8 nu=3 md=2 Nr=Nh=9 (1023 nodes), |u|<=0.5, lambda0=0, default opts", {}
    if name == "C3":
        return [lti(P.linear_chain(2, 11, 11))], "linear_chain nx=8 nu=3 md=2 Nr=Nh=11 (4095 nodes), |u|<=0.5", {}
    if name == "C1":
        return [lti(P.spring_mass())], "spring_mass example data md=3 Nr=2 Nh=10 (85 nodes)", {}
    if name == "C4":
        f = P.random_clipping_qp()
        return [dict(flat=f.as_dict(), lambda0=f.lambda0, qp=None, nodes=len(f.nk))], \
            "random_clipping_qp nx=20 nu=10 md=3, 8 levels (3280 nodes, dual blocks 60x60), unconstrained, opts of random_qp.c:131-133", dict(f.opts)
    if name == "C5":
        fs = [P.pruned_chain_qp(seed=7 + i) for i in range(256)]
        return [dict(flat=f.as_dict(), lambda0=f.lambda0, qp=None, nodes=len(f.nk)) for f in fs], \
            "256 pruned scenario trees (nx=8 nu=2, horizon 10, 1-3 children, <= 40 leaves, 50-310 nodes each; seeds 7..262), one batched call per step", dict(fs[0].opts)
    raise SystemExit(f"unknown workload {name}")


def cpu_baseline(items, opts, budget_s: float = 12.0):
    """Oracle timed on the host cores (bounded sample)."""
    import oracle_py as orc
    ncpu = os.cpu_count() or 1
    sample = items[:8]                      # C5: the first 8 trees of the batch
    best = {}
    for threads in sorted({1, min(ncpu, 16)}):
        o = orc.default_opts(num_threads=threads, **opts)
        t_end = time.perf_counter() + budget_s / 2
        tmin, iters, reps = float("inf"), 0, 0
        while reps < 20 and (time.perf_counter() < t_end or reps < 2):
            t, it = 0.0, 0
            for w in sample:
                s = orc.solve(w["flat"], o, w["lambda0"], traces=False)
                t += s["solver_time"]
                it += s["iter"]
            if t < tmin:
                tmin, iters = t, it
            reps += 1
        best[threads] = (iters / tmin, tmin, reps, iters)
    threads = max(best, key=lambda t: best[t][0])
    v, tmin, reps, iters = best[threads]
    detail = "; ".join(f"{t} thr: {best[t][0]:.0f} it/s" for t in sorted(best))
    return {"value": v, "unit": "newton_iter/s", "cores": threads, "kind": "port",
            "sample": f"min solver_time over {reps} passes over {len(sample)} tree(s) of the workload ({iters} Newton iterations per pass); "
                      f"CPU restatement (oracle), gcc -O3, not BLASFEO; {detail}; host has {ncpu} logical cores"}


def level_floors():
    """floors of a level step in isolation (cycles), from the committed microbenchmark output"""
    out = {}
    f = ROOT / "profiles" / "r03_v1_level_bench.txt"
    if f.exists():
        import re
        for line in f.read_text().splitlines():
            m = re.match(r"^(.*?)\s+(\d+) cycles\s*$", line)
            if m:
                out[m.group(1).strip()] = float(m.group(2))
    return out


def critical_path(g, p_levels: int, n_tiers: int, reps: int = 60, floors_apply: bool = True):
    """period of a pass = t(maxIter=2) - t(maxIter=1) (device times), against the chain of level floors.  The floors were
    measured for ONE shape (nx = 8, nu = 3, md = 2: 41-row x 16 blocks, tiers of 3 levels); for any other shape only the measured
    period and fixed cost are reported (`floors_apply` False): a floor of another shape is not a floor."""
    def med(k):
        for _ in range(10):
            g.solve(maxIter=k)
        for _ in range(reps):
            g.solve(maxIter=k)
        return float(np.median(g.device_times(reps))) * 1e6
    t1, t2 = med(1), med(2)
    fl = level_floors()
    back = fl.get("backward level (load, factor, store, schur, barrier)")
    fwd = fl.get("forward sweep of a tier (3 levels, one barrier)")
    sg = (fl.get("stage sweep, 15 nodes (+ barrier)", 0.0) + fl.get("G + H, 7 blocks (+ barrier)", 0.0))
    handover_us = 0.8                     # MI355X_MICROARCH.md, price list row handoff-1to1 (idle, 8 B .. 4 KB)
    out = {"pass_us": t2 - t1, "fixed_us": t1 - (t2 - t1), "levels": p_levels, "tiers": n_tiers,
           "note": "pass = G+H, backward sweep (one dependent block factorisation per level), forward sweep, trial sweep; "
                   "fixed = launch, state load, first sweep, last verdict, write-back"}
    if back and floors_apply:
        # hand-overs per pass: one per tier boundary on the way up; on the way down the bottom tier of a tree of three tiers or more
        # walks through tier 1 itself (round 4: tdunes_persist.hpp, p_forward_tier), i.e. one boundary less
        handovers = 2 * (n_tiers - 1) - (1 if n_tiers >= 3 else 0)
        floor = (p_levels * back + n_tiers * (fwd or 0.0) + sg) / (CLOCK_GHZ * 1e3) + handovers * handover_us
        out.update({"floor_us": floor, "achieved_over_floor": (t2 - t1) / floor,
                    "floor_terms": {"backward_level_cycles": back, "forward_tier_cycles": fwd, "stage_plus_gh_cycles": sg,
                                    "handover_us": handover_us, "handovers": handovers},
                    "floor_source": "profiles/r03_v1_level_bench.txt (tools/microbench/level_bench on MI355X), cycles at 2.4 GHz"})
    return out


def dumps(obj) -> str:
    """json.dumps without bare NaN / Infinity (not valid JSON): non-finite floats become null"""
    def clean(v):
        if isinstance(v, float):
            return v if np.isfinite(v) else None
        if isinstance(v, (np.floating,)):
            return float(v) if np.isfinite(v) else None
        if isinstance(v, (np.integer,)):
            return int(v)
        if isinstance(v, dict):
            return {k: clean(x) for k, x in v.items()}
        if isinstance(v, (list, tuple)):
            return [clean(x) for x in v]
        return v
    return json.dumps(clean(obj), allow_nan=False)


def self_launch(args):
    """`python bench.py --gpus N` without the launcher: start it as a child BEFORE any HIP / torch.cuda call and forward its line."""
    port = 29500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    r = subprocess.run(cmd, capture_output=True, text=True)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or not lines:
        sys.stderr.write(r.stdout[-2000:] + r.stderr[-4000:])
        raise SystemExit(f"bench.py --gpus {args.gpus}: the torch.distributed.run child failed (rc {r.returncode})")
    print(lines[-1], flush=True)
    raise SystemExit(0)


def pshard_child(steps: int):
    """The sharded leg of a multi-GPU run, in a process of its own per rank (see main): ONE C3 tree over WORLD_SIZE ranks inside the
    persistent launch.  gloo carries the 64-byte IPC handles of the hand-over slabs and the timing reduction; rank 0 prints the result."""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ["MASTER_PORT"] = os.environ["TREEQP_PSHARD_PORT"]
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from treeqp_amd import capi
    ndev = max(1, capi.device_count())
    dev = int(os.environ.get("TREEQP_PSHARD_DEVICE", "0")) % ndev
    c3, _, _ = make_workload("C3")
    f = c3[0]["flat"]
    m = capi.TqGpu(f["nk"], f["nx"], f["nu"], device=dev).upload(f, c3[0]["lambda0"])
    m.pshard_init(rank, world)
    handles = [None] * world
    dist.all_gather_object(handles, m.pshard_ipc_export())
    for r in range(world):
        if r != rank:
            m.pshard_ipc_connect(r, handles[r])
    for _ in range(5):
        dist.barrier()
        m.pshard_begin()
        res = m.pshard_end()
    dist.barrier()
    t0 = time.perf_counter()
    iters = 0
    for _ in range(steps):
        m.pshard_begin()
        res = m.pshard_end()
        iters += res["iter"]
    dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    # the sharded solution against a single-device solve of the same tree on this rank's device (outside the timed region): a hand-over
    # that is not seen in time ends in a time-out, one that is seen with stale data would only show here
    packs = [None] * world
    dist.all_gather_object(packs, m.pshard_pack())
    for r in range(world):
        if r != rank:
            m.pshard_unpack(r, packs[r])
    sol = m.solution()
    ref_m = capi.TqGpu(f["nk"], f["nx"], f["nu"], device=dev).upload(f, c3[0]["lambda0"])
    ref_r = ref_m.solve()
    ref = ref_m.solution()
    ref_m.close()
    diff = max(float(np.max(np.abs(sol[k] - ref[k]))) for k in ("x", "u", "lam", "mu_x", "mu_u"))
    same = (int(res["status"]), int(res["iter"]), int(res["ls_total"])) == (int(ref_r["status"]), int(ref_r["iter"]), int(ref_r["ls_total"]))
    ok = torch.tensor([1.0 if (same and diff < 1e-11) else 0.0, diff], dtype=torch.float64)
    okmin = ok.clone(); dist.all_reduce(okmin, op=dist.ReduceOp.MIN)
    okmax = ok.clone(); dist.all_reduce(okmax, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"iters": iters, "seconds": float(t.item()), "status": int(res["status"]),
                          "verified": bool(okmin[0].item() == 1.0), "max_abs_diff_vs_single_device": float(okmax[1].item())}), flush=True)
    m.close()
    dist.destroy_process_group()


def main():
    if "--pshard-child" in sys.argv:
        pshard_child(int(sys.argv[sys.argv.index("--steps") + 1]))
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="C2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-batched", action="store_true", help="skip the batched-throughput sweep and the critical-path leg (profiling runs: only the timed launches in the kernel statistics)")
    ap.add_argument("--mode", choices=["batch", "shard"], default="batch", help="N > 1: independent trees (weak) or one sharded tree (strong) as the headline")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse the multi-rank path on a box with fewer GPUs than ranks)")
    ap.add_argument("--strict-sharded", action="store_true", help="N > 1: exit with code 3 (after the JSON line has been printed) when the sharded leg failed; by default the exit code is 0 "
                                                                    "because the replicas' line is complete and valid -- `sharded_ok` at the top level of the line is the field to read")
    ap.add_argument("--trees", type=int, default=1, help="independent trees per GPU solved by one batched call per step (throughput mode; default 1 = the latency metric)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        if args.backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            # rehearsal: every rank on the devices that exist (round-robin), collectives on host tensors
            local_rank = local_rank % max(1, torch.cuda.device_count())
            torch.cuda.set_device(local_rank)
            dist.init_process_group(args.backend)
    red_dev = "cuda" if args.backend == "nccl" else "cpu"
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    from treeqp_amd import capi
    if capi.device_count() < 1:
        raise SystemExit("bench.py needs a HIP device: the tdunes hot path has no CPU fallback")
    dev = local_rank if world > 1 else -1

    items, desc, opts = make_workload(args.workload)
    shard = world > 1 and args.mode == "shard"
    if shard and (args.trees > 1 or len(items) > 1):
        raise SystemExit("--mode shard solves ONE tree")

    def mirror(w):
        f = w["flat"]
        return capi.TqGpu(f["nk"], f["nx"], f["nu"], device=dev).upload(f, w["lambda0"])

    mirrors = [mirror(w) for w in items for _ in range(args.trees)]
    g = mirrors[0]

    def solve_step():
        """One step: every tree of this rank once; returns (iterations, line-search trials, launches, first result)."""
        if len(mirrors) == 1:
            r = g.solve(**opts)
            return r["iter"], r["ls_total"], r["n_launches"], r
        rs = capi.solve_batch(mirrors, **opts)
        return sum(r["iter"] for r in rs), sum(r["ls_total"] for r in rs), sum(r["n_launches"] for r in rs), rs[0]

    def shard_setup(m):
        import torch
        idt = torch.zeros(128, dtype=torch.uint8, device="cuda")
        if rank == 0:
            idt.copy_(torch.frombuffer(bytearray(capi.shard_unique_id()), dtype=torch.uint8))
        dist.broadcast(idt, src=0)
        m.shard_init(rank, world, bytes(idt.cpu().numpy().tobytes()))

    if shard:
        shard_setup(g)

    def barrier():
        if dist is not None:
            import torch
            dist.barrier()
            torch.cuda.synchronize()

    # The timed region enqueues a solve as what it is on the persistent paths -- ONE kernel launch, nothing around it.  The HIP
    # event pair per launch that the roofline needs is recorded in a second region of the same solves right after it (the
    # pair is two more packets on the queue per solve: it belongs to the measurement, not to the hot path).
    for m in mirrors:
        m.event_timing(os.environ.get("TREEQP_BENCH_EVENTS") == "1")      # diagnostic: what the event pairs cost
    r = None
    for _ in range(args.warmup):
        r = solve_step()[3]
    for m in mirrors:
        m.device_times(1)                 # synchronises the mirror's stream
    barrier()
    t0 = time.perf_counter()
    iters = 0
    ls = 0
    launches = 0
    if len(mirrors) == 1:
        # the K steps as the reference's drivers run their NREP solves: a loop in C (tqgpu_solve_n), every solve waiting for its
        # verdict (status, iteration count) before the next starts -- the solver is timed, not this interpreter
        r, iters, ls, launches = g.solve_n(args.steps, **opts)
    else:
        # a batch per step: the same loop in C (tqgpu_solve_batch_n): every call returns when the verdict of every tree is on the host
        rs_, iters, ls, launches = capi.solve_batch_n(mirrors, args.steps, **opts)
        r = rs_[0]
    for m in mirrors:
        m.device_times(1)                 # synchronises: all K solves are complete (state written back)
    barrier()
    elapsed = time.perf_counter() - t0
    if r["status"] != 0:
        raise SystemExit(f"solver status {r['status']}")
    # roofline leg: the same solves with one HIP event pair per launch on the solver's stream
    for m in mirrors:
        m.event_timing(True)
    ev_steps = max(1, min(args.steps, 256))
    for _ in range(3):
        solve_step()
    ev_iters = 0
    for _ in range(ev_steps):
        ev_iters += solve_step()[0]
    dev_time = float(g.device_times(ev_steps).sum())

    tot_iters, tmax = float(iters), elapsed
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        n = torch.tensor([float(iters)], dtype=torch.float64, device=red_dev)
        dist.all_reduce(n, op=dist.ReduceOp.SUM)
        tmax = float(t.item())
        # batch: every rank solved its own tree; shard: all ranks worked on the same iterations
        tot_iters = float(n.item()) if not shard else float(iters)
    if shard:
        g.shard_gather_solution()

    out = None
    if rank == 0:
        n_trees = len(mirrors)
        kkt = None
        if items[0]["qp"] is not None and n_trees == 1:
            qp = items[0]["qp"]
            qp.set_solution(g.solution())
            kkt = qp.max_kkt_res()
        it_per_solve = iters / args.steps / n_trees
        ls_per_iter = ls / max(iters, 1)
        # algorithmic bytes: bytes per Newton iteration (closed form, per tree) x the iterations of the launch
        n_ls = max(1, round(ls_per_iter))
        if len(mirrors) > 1:
            # a batch of trees per step (C5, or --trees B): ONE launch per step carries them all; its duration is taken from the wall
            # clock of the step (members of a batch launch record no event pair of their own)
            rs = capi.solve_batch(mirrors, **opts)
            costs = [m.iteration_cost(n_ls) for m in mirrors]
            bytes_step = float(sum(b * rr["iter"] for (b, _), rr in zip(costs, rs)))
            bytes_it = float(np.mean([b for b, _ in costs]))
            flops_it = float(np.mean([f for _, f in costs]))
            launch_s = tmax / args.steps
            dev_time = launch_s * ev_steps                             # device time per iteration from the same wall clock (all trees of the step)
        else:
            bytes_it, flops_it = g.iteration_cost(n_ls)
            bytes_step = bytes_it * it_per_solve                   # mirror 0's launch: one tree
            launch_s = dev_time / ev_steps                         # HIP events on mirror 0's stream around its launch
        achieved = bytes_step / launch_s / 1e9
        traffic = None
        tf = ROOT / "profiles" / f"traffic_{args.workload}.json"
        if tf.exists() and args.trees == 1:
            traffic = json.loads(tf.read_text()).get("bytes_per_launch")        # from the committed PMC passes
        kernel = {2: "f_persist / f_mpersist: the whole solve in one launch (first sweep + all Newton iterations)",
                  3: "g_persist(_batch): the whole solve in one launch of one workgroup per tree",
                  1: "one Newton iteration = f_back x tiers, f_top, f_fwd x tiers, f_stage, k_ls_decide (tiered path)",
                  0: "blocks of 16 < d <= 64 rows: one Newton iteration = k_sgp (stage + gradient + Armijo / termination tails), k_hf_w (H + backward sweep with panel look-ahead, MFMA), k_fwd3 (forward sweep + direction test); other trees on this path: k_grad, k_check, k_hess, k_factor_all, k_forward_all, k_ls_*, k_stage"}[g.path]
        if len(mirrors) > 1 and launches <= 1.5 * args.steps * len(mirrors):
            # every tree of the step in ONE launch: the batch kernels (a member's own path, taken when it is solved alone, does not matter here)
            kernel = ("f_persist_batch: one launch per step, every tree its own set of tier-subtree workgroups" if g.path == 2 else
                      "g_persist_batch: one launch per step, one workgroup per tree (all phases of all Newton iterations of the tree)")
        out = {
            "metric": "dual_newton_iterations_per_second",
            "value": tot_iters / tmax,
            "unit": "newton_iter/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * tmax / args.steps,
            "higher_is_better": True,
            "scaling": "strong" if shard else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: {desc}", "nodes": int(sum(w["nodes"] for w in items)), "newton_iter_per_solve": it_per_solve,
                       "ls_trials_per_iter": ls_per_iter, "ms_per_newton_iter": 1e3 * tmax / max(iters, 1),
                       "device_ms_per_newton_iter": 1e3 * dev_time / max(ev_iters, 1),
                       "kernel_launches_per_solve": launches / args.steps / n_trees, "max_kkt_residual": kkt, "trees_per_gpu": n_trees,
                       "device_path": int(g.path),
                       "parallelism": ("one tree sharded by subtrees, 2 RCCL all-gathers per Newton iteration" if shard else
                                       "1 tree per GPU (independent scenario trees), no collective") if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "kernel": kernel,
                         "traffic_note": "memory-side bytes per launch (FETCH_SIZE raw + WRITE_SIZE) from profiles/traffic_<workload>.json; one launch = one solve",
                         "launch_us": 1e6 * launch_s, "launch_us_source": ("wall clock of a step (one batch launch carries every tree of the step)" if len(mirrors) > 1 else f"HIP event pair per solve on the solver's stream, mean over {ev_steps} solves of the same workload right after the timed region (the timed region itself enqueues the bare launches)"),
                         "algorithmic_bytes_per_launch": bytes_step,
                         "algorithmic_bytes_per_iteration": bytes_it, "algorithmic_flops_per_iteration": flops_it,
                         "note": "latency-bound: a chain of dependent block factorisations per tree level; on the persistent paths state and constants are LDS-resident, so memory traffic is far below the algorithmic bytes"},
        }
        if world == 1 and n_trees == 1 and g.path == 2 and not args.no_batched:
            # the dependent chain that bounds a solve, against its measured floors
            geo = g.geometry()
            f0 = items[0]["flat"]
            c2_shape = int(f0["nx"][0]) == 8 and int(f0["nu"][0]) == 3 and int(f0["nk"][0]) == 2 and len(set(map(int, f0["nk"][f0["nk"] > 0]))) == 1
            out["critical_path"] = critical_path(g, geo["levels"], geo["tiers"], floors_apply=c2_shape)
            if not c2_shape:
                out["critical_path"]["floor_note"] = "level floors exist for the (8, 3, 2) shape only (tools/microbench/level_bench); not applied to this shape"
            # throughput leg (reported beside the latency metric, never as `value`): independent trees of the same workload solved by
            # one batched call per step -- what a scenario sweep (fault_tolerance.c:486-530) gets -- swept up to what is co-resident
            cap = {"workgroups_per_tree": geo["workgroups"], "capacity": geo["capacity"], "compute_units": geo["compute_units"]}
            g.event_timing(False)              # (the roofline leg above switched the event pair on; a batch call of one tree would record it)
            sweep = []
            more = []
            nb = 1
            while True:
                while len(more) < nb - 1:
                    more.append(mirror(items[0]))
                batch = [g] + more[:nb - 1]
                for _ in range(5):
                    capi.solve_batch(batch)
                ksteps = max(20, min(args.steps, 100))
                tb0 = time.perf_counter()
                nit = 0
                nit = capi.solve_batch_n(batch, ksteps)[1]                     # (the loop in C: the library is timed, not this interpreter's ctypes marshalling of 7 x 8 result fields per step)
                g.device_times(1)                                            # synchronises
                tb = time.perf_counter() - tb0
                sweep.append({"trees_per_gpu": nb, "value": nit / tb, "ms_per_step": 1e3 * tb / ksteps})
                fit = min(64, cap["capacity"] // cap["workgroups_per_tree"])      # the largest batch that is co-resident closes the sweep
                nxt = nb + 1 if nb < 4 else nb + max(1, nb // 3)
                if nb >= fit:
                    break
                nxt = min(nxt, fit)
                nb = nxt
            best = max(sweep, key=lambda e: e["value"])
            out["batched"] = {"trees_per_gpu": best["trees_per_gpu"], "value": best["value"], "unit": "newton_iter/s", "ms_per_step": best["ms_per_step"],
                              "roofline_frac": best["value"] * bytes_it / 1e9 / HBM_PEAK_GBS, "sweep": sweep,
                              "capacity": cap, "note": "independent trees per GPU solved by one tqgpu_solve_batch call per step: ONE launch carries all trees of a shape that has a batch kernel "
                                                       "(every workgroup of it must be resident: trees_per_gpu x workgroups_per_tree <= capacity); throughput, not the latency metric"}
            for m in more:
                m.close()
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(items, opts)
    # N > 1, batch mode: ALSO the configuration north_star names -- one C3 tree sharded by subtrees over the ranks (RCCL)
    # This leg comes LAST and under a deadline: the line of the replicas above is complete before it starts, and a communicator
    # that never comes up (the RCCL transport has not run on more than one device yet) costs the extra object, not the line.
    sharded = None
    if world > 1 and not shard and (args.backend == "nccl" or os.environ.get("TREEQP_BENCH_SHARD_ANYWAY")):      # (a gloo rehearsal puts several ranks on one device, which RCCL refuses)
        import threading
        leg_done = threading.Event()
        deadline = float(os.environ.get("TREEQP_BENCH_SHARD_DEADLINE", "180"))

        def watchdog():
            if leg_done.wait(deadline):
                return
            if rank == 0:
                out["sharded"] = {"error": f"the sharded leg did not finish within {deadline:.0f} s; abandoned"}
                out["sharded_ok"] = False
                print(dumps(out), flush=True)
            os._exit(3 if args.strict_sharded else 0)      # the main thread is stuck in a collective: no orderly teardown possible

        threading.Thread(target=watchdog, daemon=True).start()
        try:
            import torch
            c3, c3desc, _ = make_workload("C3")
            ksteps = max(10, min(args.steps, 50))

            def timed(solve_once, sync):
                for _ in range(5):
                    r3 = solve_once()
                barrier()
                ts = time.perf_counter()
                it3 = 0
                for _ in range(ksteps):
                    r3 = solve_once()
                    it3 += r3["iter"]
                sync()
                barrier()
                te = time.perf_counter() - ts
                tt = torch.tensor([te], dtype=torch.float64, device=red_dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                return r3, it3, float(tt.item())

            # (1) the MI355X-native form: the workgroups of the ONE persistent launch dealt over the ranks, hand-over words written into
            # every rank's slab through IPC-mapped peer memory (tqgpu_pshard_*): no collective, no host in the loop.  It runs in a CHILD
            # process per rank (own gloo group for the handle exchange): the kernels write into other devices' memory, and a fault there
            # -- this path has run on one device only so far -- must not take the replicas' line with it.
            family, err_p, child = None, None, None
            try:
                env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_")}      # the children's rank 0 hosts their store itself
                env["TREEQP_PSHARD_PORT"] = str(int(os.environ.get("MASTER_PORT", "29500")) + 17)
                env["TREEQP_PSHARD_DEVICE"] = str(local_rank)
                cp = subprocess.run([sys.executable, str(Path(__file__).resolve()), "--pshard-child", "--steps", str(ksteps)], env=env,
                                    capture_output=True, text=True, timeout=max(30.0, deadline - 30.0))
                lines = [l for l in cp.stdout.splitlines() if l.startswith("{")]
                if cp.returncode != 0:
                    err_p = (cp.stderr or cp.stdout)[-600:]
                elif rank == 0:
                    child = json.loads(lines[-1])
            except Exception as e:
                err_p = str(e)
            if err_p is None:
                family = "persistent launch per rank (f_persist), tagged hand-over words in peer-mapped slabs, no collective"
            # every rank takes the same branch: if any rank failed on the way, all fall back
            flag = torch.tensor([0.0 if family else 1.0], dtype=torch.float64, device=red_dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            if flag.item() > 0:
                # (2) fall-back: launch-per-tier kernels with two RCCL all-gathers per Newton iteration
                family = None
                m3 = mirror(c3[0])
                shard_setup(m3)
                r3, it3, te = timed(lambda: m3.solve(), lambda: m3.device_times(1))
                family = "launch-per-tier kernels, 2 RCCL all-gathers per Newton iteration"
                m3.close()
            if flag.item() == 0 and rank == 0:
                it3, te, r3 = child["iters"], child["seconds"], {"status": child["status"]}
            elif flag.item() == 0:
                it3, te, r3 = 0, 1.0, {"status": 0}
            sharded = {"workload": f"C3: {c3desc}", "value": it3 / te, "unit": "newton_iter/s", "steps": ksteps, "ms_per_step": 1e3 * te / ksteps,
                       "scaling": "strong", "status": int(r3["status"]), "newton_iter_per_solve": it3 / ksteps, "kernel_family": family,
                       "parallelism": f"one tree, subtrees partitioned over {world} ranks", "persistent_path_error": err_p}
            if flag.item() == 0 and rank == 0 and child is not None:
                # the child compared the collected solution with a single-device solve of the same tree (outside its timed region)
                sharded["verified_against_single_device"] = bool(child.get("verified", False))
                sharded["max_abs_diff_vs_single_device"] = child.get("max_abs_diff_vs_single_device")
                if not sharded["verified_against_single_device"]:
                    sharded["error"] = "the sharded solve's solution or counts differ from the single-device solve"
        except Exception as e:          # the replica line must not be lost to a failure of the extra leg
            sharded = {"error": str(e)}
        leg_done.set()

    if rank == 0:
        if sharded is not None:
            out["sharded"] = sharded
            out["sharded_ok"] = "error" not in sharded          # top level: a failed sharded leg must not read as success
        print(dumps(out), flush=True)
    if sharded is not None and "error" in sharded:
        os._exit(3 if args.strict_sharded else 0)          # the other ranks may be stuck in a collective of the failed leg (their deadline ends them): no orderly teardown with them
    for m in mirrors:
        m.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
