"""Shard-by-query work distribution over the GPUs of one node.

Alignments of different pairs are independent, so there is no data-path collective: the only traffic is a
broadcast of the job description (model type + parameters, a few kB), a scatter of pair shards and a gather
of results — `torch.distributed` over RCCL/xGMI on the GPUs (backend "nccl"), gloo in the CPU tests.
Results are returned in submission order, the order `GAM_Result_submit` prints them (gam.c:1252).

`align` is whatever computes a shard: the GPU engine in production
(`lambda model_type, pairs: Engine(local_rank).find_path(Model(model_type), pairs)`), the oracle in the CPU
test of this plumbing.
"""
import torch.distributed as dist


def shard_bounds(n_items, world_size):
    """Contiguous, balanced shards; rank r owns [bounds[r], bounds[r+1])."""
    base, extra = divmod(n_items, world_size)
    bounds = [0]
    for r in range(world_size):
        bounds.append(bounds[-1] + base + (1 if r < extra else 0))
    return bounds


def shard_by_cost(costs, world_size):
    """Longest-first round-robin for uneven pairs (cells = (Q+1)(T+1)); returns a list of index lists."""
    order = sorted(range(len(costs)), key=lambda i: -costs[i])
    shards = [[] for _ in range(world_size)]
    load = [0] * world_size
    for i in order:
        r = min(range(world_size), key=lambda x: load[x])
        shards[r].append(i)
        load[r] += costs[i]
    return [sorted(s) for s in shards]


def distributed_find_path(align, model_type, pairs, src=0):
    """Rank `src` holds `pairs`; every rank returns the full result list in submission order."""
    rank, world = dist.get_rank(), dist.get_world_size()
    header = [model_type, len(pairs) if rank == src else 0]
    dist.broadcast_object_list(header, src=src)                 # work-queue broadcast
    model_type, n = header
    if rank == src:
        costs = [(len(q) + 1) * (len(t) + 1) for q, t in pairs]
        index_shards = shard_by_cost(costs, world)
        payload = [[(i, pairs[i]) for i in s] for s in index_shards]
    else:
        payload = [None] * world
    mine = [None]
    dist.scatter_object_list(mine, payload, src=src)            # pair shards
    local = mine[0]
    results = align(model_type, [p for _, p in local])
    gathered = [None] * world
    dist.all_gather_object(gathered, [(i, r) for (i, _), r in zip(local, results)])   # result gather
    out = [None] * n
    for shard in gathered:
        for i, r in shard:
            out[i] = r
    return out
