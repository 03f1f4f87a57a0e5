"""Worker of tests/test_gpu_pshard.py::test_processes_share_one_tree_through_ipc_slabs: rank `r` of `n` PROCESSES on one GPU.
Each process creates a mirror of the whole tree, becomes a rank of the sharded persistent solve, exchanges the IPC handles of
the hand-over slabs and -- afterwards -- its share of the solution through torch.distributed (gloo: host-staged; RCCL refuses
two ranks on one device), and checks the collected solution against an unsharded solve of its own."""
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    rank, n, port, levels = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=n)
    from treeqp_amd import capi, problems as P
    p = P.linear_chain(2, levels, levels)
    nk = p.nk()
    nx = np.full(p.Nn, p.nx, dtype=np.int32)
    nu = np.where(nk > 0, p.nu, 0).astype(np.int32)
    flat = capi.TreeQp(nx, nu, nk).fill_lti(p).flat()
    ref_m = capi.TqGpu(nk, nx, nu).upload(flat, p.lambda0)
    ref_r = ref_m.solve()
    ref = ref_m.solution()
    ref_m.close()
    g = capi.TqGpu(nk, nx, nu).upload(flat, p.lambda0)
    g.pshard_init(rank, n)
    handles = [None] * n
    dist.all_gather_object(handles, g.pshard_ipc_export())
    for r in range(n):
        if r != rank:
            g.pshard_ipc_connect(r, handles[r])
    results = []
    for rep in range(3):                                   # several solves: the launch numbers advance in step on every rank
        dist.barrier()                                     # every rank's launch goes out now: they wait for each other inside the kernels
        g.pshard_begin()
        res = g.pshard_end()
        results.append((res["status"], res["iter"], res["ls_total"]))
    packs = [None] * n
    dist.all_gather_object(packs, g.pshard_pack())
    for r in range(n):
        if r != rank:
            g.pshard_unpack(r, packs[r])
    sol = g.solution()
    g.close()
    assert all(t == (ref_r["status"], ref_r["iter"], ref_r["ls_total"]) for t in results), (results, ref_r)
    err = max(float(np.max(np.abs(sol[k] - ref[k]))) for k in ("x", "u", "lam", "mu_x", "mu_u"))
    assert err < 1e-11, err
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank}: ok, {results[0]}, max |sharded - single device| = {err:.2e}", flush=True)


if __name__ == "__main__":
    main()
