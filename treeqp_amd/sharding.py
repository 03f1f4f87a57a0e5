"""Partition plan of the sharded (one tree over several GPUs) mode -- pure Python restatement of
``shard_build_lists`` in csrc/device/tdunes_device.hip, used by bench.py for reporting and by the
CPU (gloo) protocol tests.  Uniform complete md-ary trees only (the fused path)."""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np


def first_of_level(md: int, level: int) -> int:
    return (md ** level - 1) // (md - 1)


def tier_height(md: int, waves: int = 4) -> int:
    th = 1
    while md ** th <= waves:
        th += 1
    return th


@dataclass
class ShardPlan:
    md: int
    Nh: int
    nranks: int
    rank: int
    tiers: list            # [(l0, l1, grid)] bottom-up
    part_top: int          # highest partitioned tier
    lb: int                # boundary level: blocks/nodes of this level are the partitioned subtree roots
    owned_nodes: np.ndarray
    replicated_nodes: np.ndarray
    gh_list: np.ndarray    # blocks above tier 0 whose G+H this rank computes
    gh_counted: int
    boundary_range: tuple  # (first node, count) of this rank's boundary roots
    exchange_doubles: dict = field(default_factory=dict)


def plan(md: int, Nh: int, nranks: int, rank: int, nx: int = 8, waves: int = 4) -> ShardPlan:
    th = tier_height(md, waves)
    nt = (Nh + th - 1) // th
    tiers = []
    for i in range(nt):
        l1 = Nh - i * th
        l0 = max(0, l1 - th)
        tiers.append((l0, l1, md ** l0))
    part_top = -1
    for i in range(nt - 1):
        if tiers[i][2] % nranks == 0 and tiers[i][2] >= nranks:
            part_top = i
    if part_top < 0:
        raise ValueError("tree too small to shard over this many ranks")
    lb, l00 = tiers[part_top][0], tiers[0][0]
    owned, gh = [], []
    for l in range(lb, Nh + 1):
        w = md ** l // nranks
        f0 = first_of_level(md, l) + rank * w
        owned += list(range(f0, f0 + w))
        if l < l00:
            gh += list(range(f0, f0 + w))
    gh_counted = len(gh)
    repl = list(range(0, first_of_level(md, lb)))
    gh += repl
    if rank == 0:
        gh_counted = len(gh)
    w = md ** lb // nranks
    sch = nx * nx + nx
    return ShardPlan(md=md, Nh=Nh, nranks=nranks, rank=rank, tiers=tiers, part_top=part_top, lb=lb,
                     owned_nodes=np.asarray(owned), replicated_nodes=np.asarray(repl), gh_list=np.asarray(gh),
                     gh_counted=gh_counted, boundary_range=(first_of_level(md, lb) + rank * w, w),
                     exchange_doubles={"exchange1": nranks * (w * sch + 1), "exchange2": nranks * (2 + 2 * w * nx)})


def pshard_plan(md: int, Nh: int, nranks: int, rank: int, waves: int = 4) -> dict:
    """Partition of the PERSISTENT launch's workgroups over the ranks of a sharded solve (restatement of tqgpu_pshard_plan):
    workgroups are numbered tier by tier from the bottom, one per tier subtree; tiers whose subtree count is a multiple of nranks
    go to the ranks by contiguous subtree ranges, the tiers above them to rank 0 -- no workgroup exists twice."""
    th = tier_height(md, waves)
    nt = (Nh + th - 1) // th
    tiers = []
    for i in range(nt):
        l1 = Nh - i * th
        l0 = max(0, l1 - th)
        tiers.append((l0, l1, md ** l0))
    top = nt - 1 if nranks == 1 else -1
    if nranks > 1:
        for i in range(nt - 1):
            if tiers[i][2] % nranks == 0 and tiers[i][2] >= nranks:
                top = i
    if top < 0:
        raise ValueError("tree too small to shard over this many ranks")
    wgs, wg0 = [], 0
    for i, (_, _, grid) in enumerate(tiers):
        for q in range(grid):
            owner = q // (grid // nranks) if (nranks > 1 and i <= top) else 0
            if owner == rank:
                wgs.append(wg0 + q)
        wg0 += grid
    lb = tiers[top][0] if nranks > 1 else 0
    # what a rank holds of the solution afterwards: node data of level l >= lb by contiguous chunks, below that rank 0;
    # the duals of a node's own edge belong to the block of its parent: chunks from level lb + 1 on
    def chunk(l, first_level):
        w = md ** l
        f0 = first_of_level(md, l)
        if l >= first_level and nranks > 1:
            c = w // nranks
            return (f0 + rank * c, c)
        return (f0, w) if rank == 0 else (f0, 0)
    return {"wgs": np.asarray(wgs, dtype=np.int64), "total": wg0, "part_top": top if nranks > 1 else -1, "boundary_level": lb,
            "node_chunks": [chunk(l, lb) for l in range(Nh + 1)], "dual_chunks": [chunk(l, lb + 1) for l in range(Nh + 1)]}

