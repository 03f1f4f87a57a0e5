#!/usr/bin/env python3
"""bench.py -- dual-Newton iterations/s of the tdunes hot path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line on rank 0.
For N > 1 it is launched under torch.distributed.run (one rank per GPU, RCCL).

* step      = one full tdunes solve of the workload from the same resident lambda0 (the reference's
              own timing protocol: examples/spring_mass_dual_newton_tree.c:135-140); QP data,
              index tables and lambda0 are resident in HBM before the timed region starts.
* value     = Newton iterations of ALL ranks / wall time of the K steps (max over ranks).
* workload  = BASELINE.json configs[1] ("C2"): linear-chain spring-mass tree nx=8, nu=3, 10 levels,
              branching 2 -> 1023 nodes.  N > 1, default `--mode batch`: one such tree per
              GPU (independent scenario trees, weak scaling, no data-path collective).  `--mode shard`:
              ONE tree, subtrees partitioned over the ranks, two small RCCL all-gathers per Newton
              iteration (strong scaling, latency-bound by construction).
* roofline  = algorithmic bytes of the Newton iterations (closed form of SURVEY.md §8d, evaluated
              by tqgpu_iteration_cost) / device time between HIP events recorded on the solver's own
              stream around each solve (tqgpu_get_device_times), against the 8 TB/s HBM3E peak.
* cpu_baseline = the CPU oracle ("port": restatement of the reference algorithm, NOT BLASFEO
              HIGH_PERFORMANCE) on the same workload, min over repetitions, rank 0 at N=1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
for p in (ROOT, ROOT / "oracle"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)


def make_workload(name: str):
    from treeqp_amd import problems as P
    if name == "C2":
        return P.linear_chain(2, 9, 9), "linear_chain nx=8 nu=3 md=2 Nr=Nh=9 (1023 nodes), |u|<=0.5, lambda0=0, default opts"
    if name == "C3":
        return P.linear_chain(2, 11, 11), "linear_chain nx=8 nu=3 md=2 Nr=Nh=11 (4095 nodes), |u|<=0.5"
    if name == "C1":
        return P.spring_mass(), "spring_mass example data md=3 Nr=2 Nh=10 (85 nodes)"
    raise SystemExit(f"unknown workload {name}")


def cpu_baseline(p, flat, budget_s: float = 12.0):
    """Oracle timed on the host cores (bounded sample)."""
    import oracle_py as orc
    ncpu = os.cpu_count() or 1
    best = {}
    for threads in sorted({1, min(ncpu, 16)}):
        o = orc.default_opts(num_threads=threads)
        t_end = time.perf_counter() + budget_s / 2
        tmin, iters, reps = float("inf"), 0, 0
        while reps < 20 and (time.perf_counter() < t_end or reps < 3):
            s = orc.solve(flat, o, p.lambda0, traces=False)
            tmin = min(tmin, s["solver_time"])
            iters = s["iter"]
            reps += 1
        best[threads] = (iters / tmin, tmin, reps, iters)
    threads = max(best, key=lambda t: best[t][0])
    v, tmin, reps, iters = best[threads]
    detail = "; ".join(f"{t} thr: {best[t][0]:.0f} it/s" for t in sorted(best))
    return {"value": v, "unit": "newton_iter/s", "cores": threads, "kind": "port",
            "sample": f"min solver_time over {reps} full solves ({iters} Newton iterations each) of the same workload; "
                      f"CPU restatement (oracle), gcc -O3, not BLASFEO; {detail}; host has {ncpu} logical cores"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="C2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-batched", action="store_true", help="skip the batched-throughput leg (profiling runs: only the timed launches in the kernel statistics)")
    ap.add_argument("--mode", choices=["batch", "shard"], default="batch", help="N > 1: independent trees (weak) or one sharded tree (strong)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse the multi-rank path on a box with fewer GPUs than ranks)")
    ap.add_argument("--trees", type=int, default=1, help="independent trees per GPU solved by one batched call per step (throughput mode; default 1 = the latency metric)")
    args = ap.parse_args()

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
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    from treeqp_amd import capi
    if capi.device_count() < 1:
        raise SystemExit("bench.py needs a HIP device: the tdunes hot path has no CPU fallback")

    p, desc = make_workload(args.workload)
    nk = p.nk()
    nx = np.full(p.Nn, p.nx, dtype=np.int32)
    nu = np.where(nk > 0, p.nu, 0).astype(np.int32)
    # QP built through the reference-compatible host API, then made resident through the C-ABI
    qp = capi.TreeQp(nx, nu, nk).fill_lti(p)
    flat = qp.flat()
    g = capi.TqGpu(nk, nx, nu, device=local_rank if world > 1 else -1).upload(flat, p.lambda0)
    extra = [capi.TqGpu(nk, nx, nu, device=local_rank if world > 1 else -1).upload(flat, p.lambda0) for _ in range(max(0, args.trees - 1))]
    mirrors = [g] + extra
    shard = world > 1 and args.mode == "shard"
    if shard and extra:
        raise SystemExit("--trees > 1 is a batch-mode option")

    def solve_step():
        """One step: every tree of this rank once; returns (iterations, line-search trials, launches, last result)."""
        if not extra:
            r = g.solve()
            return r["iter"], r["ls_total"], r["n_launches"], r
        rs = capi.solve_batch(mirrors)
        return sum(r["iter"] for r in rs), sum(r["ls_total"] for r in rs), sum(r["n_launches"] for r in rs), rs[0]
    if shard:
        import torch
        idt = torch.zeros(128, dtype=torch.uint8, device="cuda")
        if rank == 0:
            idt.copy_(torch.frombuffer(bytearray(capi.shard_unique_id()), dtype=torch.uint8))
        dist.broadcast(idt, src=0)
        g.shard_init(rank, world, bytes(idt.cpu().numpy().tobytes()))

    def barrier():
        if dist is not None:
            import torch
            dist.barrier()
            torch.cuda.synchronize()

    r = None
    for _ in range(args.warmup):
        r = solve_step()[3]
    barrier()
    t0 = time.perf_counter()
    dev_time = 0.0
    iters = 0
    ls = 0
    launches = 0
    pending = 0
    for _ in range(args.steps):
        it_, ls_, la_, r = solve_step()   # returns when the verdict (status, iteration count) of every tree is on the host
        iters += it_
        ls += ls_
        launches += la_
        pending += 1
        if pending == 256:          # HIP-event times of the solves, fetched in batches (each fetch synchronises the stream)
            dev_time += float(g.device_times(pending).sum())
            pending = 0
    dev_time += float(g.device_times(pending).sum()) if pending else 0.0      # synchronises: all K solves are complete
    barrier()
    elapsed = time.perf_counter() - t0
    if r["status"] != 0:
        raise SystemExit(f"solver status {r['status']}")

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

    if rank == 0:
        sol = g.solution()
        import ctypes as C
        qp.set_solution(sol)
        kkt = qp.max_kkt_res()
        it_per_solve = iters / args.steps / len(mirrors)
        ls_per_iter = ls / max(iters, 1)
        bytes_it, flops_it = g.iteration_cost(max(1, round(ls_per_iter)))
        # dominant kernel: one f_persist launch = one solve (persistent path); algorithmic bytes per launch =
        # closed-form bytes per Newton iteration x the iterations of the launch, over the launch's duration
        achieved = bytes_it * (iters / len(mirrors)) / dev_time / 1e9          # one tree's launches (mirror 0)
        traffic = None
        tf = ROOT / "profiles" / f"traffic_{args.workload}.json"
        if tf.exists() and g.path == 2:
            traffic = json.loads(tf.read_text()).get("bytes_per_launch")        # from the committed PMC passes
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
            "config": {"workload": f"{args.workload}: {desc}", "nodes": int(p.Nn), "newton_iter_per_solve": it_per_solve,
                       "ls_trials_per_iter": ls_per_iter, "ms_per_newton_iter": 1e3 * tmax / max(iters, 1),
                       "device_ms_per_newton_iter": 1e3 * dev_time / max(iters, 1),
                       "kernel_launches_per_solve": launches / args.steps / len(mirrors), "max_kkt_residual": kkt, "trees_per_gpu": len(mirrors),
                       "parallelism": ("one tree sharded by subtrees, 2 RCCL all-gathers per Newton iteration" if shard else
                                       "1 tree per GPU (independent scenario trees), no collective") if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": ("f_persist: the whole solve in one launch (first sweep + all Newton iterations)" if g.path == 2 else "one Newton iteration = f_back x tiers, f_top, f_fwd x tiers, f_stage, k_ls_decide (tiered path)") if g.fused else
                                   "one Newton iteration = k_grad,k_check,k_hess,k_factor x levels,k_forward x levels,k_ls_*,k_stage (generic path)",
                         "traffic_note": "memory-side bytes per launch (FETCH_SIZE raw + WRITE_SIZE) from profiles/traffic_<workload>.json; one launch = one solve",
                         "launch_us": 1e6 * dev_time / args.steps, "algorithmic_bytes_per_launch": bytes_it * it_per_solve,
                         "algorithmic_bytes_per_iteration": bytes_it, "algorithmic_flops_per_iteration": flops_it,
                         "note": "latency-bound: a chain of dependent 25x16 block factorisations per tree level; the solve's state is LDS-resident, so memory traffic is far below the algorithmic bytes"},
        }
        if world == 1 and args.trees == 1 and g.path == 2 and not args.no_batched:
            # throughput leg (reported beside the latency metric, never as `value`): independent trees of the same
            # workload solved by one batched call per step -- what a scenario sweep (fault_tolerance.c:486-530) gets
            nb = 3
            more = [capi.TqGpu(nk, nx, nu).upload(flat, p.lambda0) for _ in range(nb - 1)]
            batch = [g] + more
            for _ in range(10):
                capi.solve_batch(batch)
            tb0 = time.perf_counter()
            nit = 0
            ksteps = max(20, min(args.steps, 200))
            for _ in range(ksteps):
                nit += sum(rr["iter"] for rr in capi.solve_batch(batch))
            g.device_times(1)                                            # synchronises
            tb = time.perf_counter() - tb0
            out["batched"] = {"trees_per_gpu": nb, "value": nit / tb, "unit": "newton_iter/s", "steps": ksteps, "ms_per_step": 1e3 * tb / ksteps,
                              "note": "independent trees per GPU, one persistent launch each, concurrently resident; throughput, not the latency metric"}
            for m in more:
                m.close()
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(p, flat)
        print(json.dumps(out), flush=True)
    for m in mirrors:
        m.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
