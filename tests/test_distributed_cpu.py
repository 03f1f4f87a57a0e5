"""N > 1 paths on CPU (gloo, world_size 2): the sharded-mode exchange protocol and the bench harness'
reductions.  The solver itself has no CPU path; these tests cover what runs on the host side of the
multi-GPU modes: the partition plan (treeqp_amd/sharding.py mirrors shard_build_lists), the in-place
rank-ordered all-gathers and the "identical decision on every rank" property."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from treeqp_amd import sharding


def test_plan_covers_every_node_exactly_once():
    for md, Nh, n in [(2, 9, 2), (2, 9, 4), (2, 9, 8), (2, 11, 8), (2, 6, 8), (3, 4, 3)]:
        plans = [sharding.plan(md, Nh, n, r) for r in range(n)]
        Nn = sharding.first_of_level(md, Nh + 1)
        owned = np.concatenate([p.owned_nodes for p in plans])
        repl = plans[0].replicated_nodes
        assert len(np.unique(owned)) == len(owned)                      # no node owned twice
        assert np.array_equal(np.sort(np.concatenate([owned, repl])), np.arange(Nn))
        lb = plans[0].lb
        f0, gb = sharding.first_of_level(md, lb), md ** lb
        starts = [p.boundary_range[0] for p in plans]
        assert starts == [f0 + r * gb // n for r in range(n)]           # rank-ordered contiguous ranges
        # replicated G+H blocks are counted for the termination norm by rank 0 only
        assert plans[0].gh_counted == len(plans[0].gh_list)
        assert all(p.gh_counted == len(p.gh_list) - len(repl) for p in plans[1:])
        # C2 / 8 ranks: one 72-double Schur record per rank (SURVEY §8e "nx^2 + nx doubles")
    p8 = sharding.plan(2, 9, 8, 3)
    assert p8.lb == 3 and p8.boundary_range == (7 + 3, 1) and p8.exchange_doubles["exchange1"] == 8 * 73


def test_product_planner_is_the_python_planner(capi):
    """shard_build_lists (the product, C) and treeqp_amd/sharding.py (the restatement the gloo tests exercise) cannot drift apart:
    the product's host-only planner, exported as tqgpu_shard_plan, gives the same plan for every rank of every case."""
    import ctypes as C
    L = capi.lib()
    for md, Nh, n in [(2, 9, 1), (2, 9, 2), (2, 9, 4), (2, 9, 8), (2, 11, 2), (2, 11, 4), (2, 11, 8), (2, 6, 8), (3, 4, 3), (4, 4, 4)]:
        for r in range(n):
            pl = sharding.plan(md, Nh, n, r)
            v = [C.c_int() for _ in range(5)]          # part_top, boundary level, gh_counted, gh_n, owned_n
            assert L.tqgpu_shard_plan(md, 8, Nh, n, r, C.byref(v[0]), C.byref(v[1]), C.byref(v[2]), None, 0, C.byref(v[3]), None, 0, C.byref(v[4])) == 0
            gh = (C.c_int * max(1, v[3].value))()
            own = (C.c_int * max(1, v[4].value))()
            assert L.tqgpu_shard_plan(md, 8, Nh, n, r, None, None, None, gh, v[3].value, None, own, v[4].value, None) == 0
            assert (v[0].value, v[1].value, v[2].value) == (pl.part_top, pl.lb, pl.gh_counted), (md, Nh, n, r)
            assert list(gh)[:v[3].value] == list(pl.gh_list)
            assert list(own)[:v[4].value] == list(pl.owned_nodes)
    assert L.tqgpu_shard_plan(2, 8, 3, 64, 0, None, None, None, None, 0, None, None, 0, None) != 0      # too many ranks


def test_plan_rejects_too_many_ranks():
    with pytest.raises(ValueError):
        sharding.plan(2, 3, 64, 0)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        md, Nh, nx = 2, 9, 8
        pl = sharding.plan(md, Nh, world, rank, nx)
        Nn = sharding.first_of_level(md, Nh + 1)
        sch = nx * nx + nx
        # exchange 1: in-place all-gather of the boundary Schur records (rank r fills its own range)
        f0, w = pl.boundary_range
        fb, gb = sharding.first_of_level(md, pl.lb), md ** pl.lb
        sbuf = torch.full((Nn * sch,), float("nan"), dtype=torch.float64)
        sbuf[f0 * sch:(f0 + w) * sch] = torch.arange(f0 * sch, (f0 + w) * sch, dtype=torch.float64)
        region = sbuf[fb * sch:(fb + gb) * sch]
        dist.all_gather_into_tensor(region, region[rank * w * sch:(rank + 1) * w * sch].clone())
        ok1 = bool(torch.equal(region, torch.arange(fb * sch, (fb + gb) * sch, dtype=torch.float64)))
        # termination partials: max in rank order is the global max on every rank
        xerr = torch.zeros(world, dtype=torch.float64)
        mine = torch.tensor([0.25 + rank], dtype=torch.float64)
        dist.all_gather_into_tensor(xerr, mine)
        # exchange 2: {fval, dot} partials summed in RANK ORDER -> bitwise identical decision input
        xs = torch.zeros(2 * world, dtype=torch.float64)
        part = torch.tensor([1e16 * (rank == 0) + 1.0 + rank * 1e-3, -3.0 * (rank + 1)], dtype=torch.float64)
        dist.all_gather_into_tensor(xs, part)
        f = 0.0
        d = 0.0
        for r in range(world):
            f += float(xs[2 * r])
            d += float(xs[2 * r + 1])
        # bench harness reductions (max time, summed iterations)
        t = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        n = torch.tensor([3.0 * 10], dtype=torch.float64)
        dist.all_reduce(n, op=dist.ReduceOp.SUM)
        dist.barrier()
        q.put((rank, ok1, float(xerr.max()), f.hex() if hasattr(f, "hex") else f, d, float(t), float(n)))
    finally:
        dist.destroy_process_group()


def test_gloo_world2_exchange_protocol():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, ok0, e0, f0, d0, t0, n0), (r1, ok1, e1, f1, d1, t1, n1) = out
    assert ok0 and ok1                          # both ranks hold every boundary record after the gather
    assert e0 == e1 == 1.25                     # same termination norm everywhere
    assert f0 == f1 and d0 == d1                # rank-ordered sums: bitwise identical on both ranks
    assert t0 == t1 == 2.0 and n0 == n1 == 60.0  # bench: max time over ranks, summed iterations


# ---- sharded mode INSIDE the persistent launch (tqgpu_pshard_*): host side ------------------------------------------------

def test_pshard_plan_deals_every_workgroup_exactly_once(capi):
    """The product's planner (tqgpu_pshard_plan, host arithmetic, used by tqgpu_pshard_init) against its Python restatement; over
    the ranks every workgroup of the single-device launch appears exactly once, partitioned tiers in equal contiguous ranges."""
    import ctypes as C
    L = capi.lib()
    for md, Nh, n in [(2, 9, 1), (2, 9, 2), (2, 9, 4), (2, 9, 8), (2, 11, 2), (2, 11, 4), (2, 11, 8), (2, 6, 8), (3, 4, 3), (4, 4, 4), (2, 14, 8)]:
        seen = []
        for r in range(n):
            pl = sharding.pshard_plan(md, Nh, n, r)
            cnt, top, lb = C.c_int(), C.c_int(), C.c_int()
            assert L.tqgpu_pshard_plan(md, Nh, n, r, None, 0, C.byref(cnt), C.byref(top), C.byref(lb)) == 0
            wgs = (C.c_int * max(1, cnt.value))()
            assert L.tqgpu_pshard_plan(md, Nh, n, r, wgs, cnt.value, C.byref(cnt), None, None) == 0
            assert list(wgs)[:cnt.value] == list(pl["wgs"]) and (top.value, lb.value) == (pl["part_top"], pl["boundary_level"]), (md, Nh, n, r)
            seen += list(pl["wgs"])
            # every rank's node / dual chunks partition each level
        assert sorted(seen) == list(range(sharding.pshard_plan(md, Nh, n, 0)["total"])), (md, Nh, n)
        for l in range(Nh + 1):
            for key in ("node_chunks", "dual_chunks"):
                chunks = [sharding.pshard_plan(md, Nh, n, r)[key][l] for r in range(n)]
                cover = sorted(i for f0, c in chunks for i in range(f0, f0 + c))
                f0 = sharding.first_of_level(md, l)
                assert cover == list(range(f0, f0 + md ** l)), (md, Nh, n, l, key)
    assert L.tqgpu_pshard_plan(2, 3, 64, 0, None, 0, None, None, None) != 0          # too many ranks for the tree
    with pytest.raises(ValueError):
        sharding.pshard_plan(2, 3, 64, 0)


def _pshard_collect_worker(rank, world, port, md, Nh, q):
    """What tqgpu_pshard_pack / _unpack + an all-gather do after a sharded solve, on host arrays: every rank holds its chunks of the
    per-node and per-edge data (and rank 0 the levels above the partition); after the exchange every rank holds everything."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    Nn = sharding.first_of_level(md, Nh + 1)
    truth_x = np.arange(Nn, dtype=np.float64) * 1.5 + 0.25
    truth_l = -np.arange(Nn, dtype=np.float64) - 7.0
    pl = sharding.pshard_plan(md, Nh, world, rank)
    x = np.full(Nn, np.nan)
    lam = np.full(Nn, np.nan)
    for f0, c in pl["node_chunks"]:
        x[f0:f0 + c] = truth_x[f0:f0 + c]
    for f0, c in pl["dual_chunks"]:
        lam[f0:f0 + c] = truth_l[f0:f0 + c]
    p0 = sharding.pshard_plan(md, Nh, world, 0)
    size = sum(c for _, c in p0["node_chunks"]) + sum(c for _, c in p0["dual_chunks"])          # rank 0 holds the most: equal-sized buffers
    def pack():
        out = np.zeros(size)
        o = 0
        for (f0, c), (g0, e) in zip(pl["node_chunks"], pl["dual_chunks"]):
            out[o:o + c] = x[f0:f0 + c]; o += c
            out[o:o + e] = lam[g0:g0 + e]; o += e
        return out
    bufs = [torch.zeros(size, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(bufs, torch.from_numpy(pack()))
    for r in range(world):
        pr = sharding.pshard_plan(md, Nh, world, r)
        b, o = bufs[r].numpy(), 0
        for (f0, c), (g0, e) in zip(pr["node_chunks"], pr["dual_chunks"]):
            x[f0:f0 + c] = b[o:o + c]; o += c
            lam[g0:g0 + e] = b[o:o + e]; o += e
    q.put((rank, bool(np.array_equal(x, truth_x) and np.array_equal(lam, truth_l))))
    dist.barrier()
    dist.destroy_process_group()


def test_pshard_solution_collection_protocol_world_size_2():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_pshard_collect_worker, args=(r, 2, port, 2, 9, q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p_ in procs:
        p_.join(timeout=60)
    assert res == [(0, True), (1, True)]

