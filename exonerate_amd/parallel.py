"""Shard-by-query work distribution over the GPUs of one node.

Alignments of different pairs are independent, so there is no data-path collective: the only traffic is a
broadcast of the job header (model type, counts), a scatter of pair shards (residues as one uint8 tensor per rank) and a
gather of results (score, region, operations as one int32 tensor per rank) — `torch.distributed` collectives on TENSORS:
over RCCL/xGMI on the GPUs (backend "nccl": tensors live on the rank's device), gloo in the CPU tests (same code, CPU
tensors).  Results come back in submission order, the order `GAM_Result_submit` prints them (gam.c:1252).

`align(model_type, pairs)` is whatever computes a shard and returns, per pair, None or something with score / region /
ops (a dict with those keys or an exonerate_amd.Alignment): the GPU engine in production
(`lambda model_type, pairs: Engine(local_rank).find_path(Model(model_type), pairs)`), the oracle in the CPU test of this
plumbing.
"""
import torch
import torch.distributed as dist

MODEL_NAME_BYTES = 64


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


def collective_device():
    """Where the tensors of a collective live: the rank's GPU under RCCL, the host under gloo."""
    if dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def _as_bytes(s):
    return s if isinstance(s, (bytes, bytearray)) else s.encode()


def pack_shard(indices, pairs, n_slots, n_bytes):
    """(index, query length, target length) rows and the residues of a shard, padded to the launch-wide sizes."""
    meta = torch.full((n_slots, 3), -1, dtype=torch.int64)
    blob = bytearray()
    for k, i in enumerate(indices):
        q, t = _as_bytes(pairs[i][0]), _as_bytes(pairs[i][1])
        meta[k, 0], meta[k, 1], meta[k, 2] = i, len(q), len(t)
        blob += q + t
    res = torch.zeros(n_bytes, dtype=torch.uint8)
    if blob:
        res[:len(blob)] = torch.frombuffer(bytes(blob), dtype=torch.uint8)
    return meta, res


def unpack_shard(meta, res):
    meta = meta.cpu()
    raw = res.cpu().numpy().tobytes()
    out, pos = [], 0
    for k in range(meta.shape[0]):
        i, ql, tl = int(meta[k, 0]), int(meta[k, 1]), int(meta[k, 2])
        if i < 0:
            break
        out.append((i, (raw[pos:pos + ql], raw[pos + ql:pos + ql + tl])))
        pos += ql + tl
    return out


def _parts(r):
    if r is None:
        return None
    if isinstance(r, dict):
        return r["score"], r["region"], r["ops"]
    return r.score, r.region, r.ops


def encode_results(indexed):
    """[(index, result)] -> one int32 tensor: index, valid, score, region (4), n_ops, (transition, length) * n_ops per pair."""
    flat = []
    for i, r in indexed:
        p = _parts(r)
        if p is None:
            flat += [i, 0, 0, 0, 0, 0, 0, 0]
            continue
        score, region, ops = p
        flat += [i, 1, score] + [int(x) for x in region] + [len(ops)]
        for t, l in ops:
            flat += [int(t), int(l)]
    return torch.tensor(flat, dtype=torch.int32) if flat else torch.zeros(0, dtype=torch.int32)


def decode_results(flat):
    """The inverse: [(index, None | {"score", "region", "ops"})]."""
    v = flat.cpu().tolist()
    out, pos = [], 0
    while pos < len(v):
        i, valid, score, qs, ts, ql, tl, n = v[pos:pos + 8]
        pos += 8
        if not valid:
            out.append((i, None))
            continue
        ops = [[v[pos + 2 * k], v[pos + 2 * k + 1]] for k in range(n)]
        pos += 2 * n
        out.append((i, {"score": score, "region": [qs, ts, ql, tl], "ops": ops}))
    return out


def all_gather_ragged(t, dev):
    """all_gather of 1-D tensors of different lengths: sizes first, then the payloads padded to the longest."""
    world = dist.get_world_size()
    size = torch.tensor([t.numel()], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, size)
    longest = max(1, max(int(s.item()) for s in sizes))
    mine = torch.zeros(longest, dtype=t.dtype, device=dev)
    mine[:t.numel()] = t.to(dev)
    got = [torch.zeros(longest, dtype=t.dtype, device=dev) for _ in range(world)]
    dist.all_gather(got, mine)
    return [g[:int(s.item())] for g, s in zip(got, sizes)]


def distributed_find_path(align, model_type, pairs, src=0):
    """Rank `src` holds `pairs`; every rank returns the full result list in submission order: per pair None or
    {"score", "region", "ops"} (Alignment.from_parts turns one back into lines of output)."""
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = collective_device()
    # work-queue broadcast: model type, pair count, shard sizes
    name = torch.zeros(MODEL_NAME_BYTES, dtype=torch.uint8)
    head = torch.zeros(3, dtype=torch.int64)
    shards = None
    if rank == src:
        raw = model_type.encode()
        assert len(raw) < MODEL_NAME_BYTES
        name[:len(raw)] = torch.frombuffer(raw, dtype=torch.uint8)
        costs = [(len(q) + 1) * (len(t) + 1) for q, t in pairs]
        shards = shard_by_cost(costs, world)
        head[0] = len(pairs)
        head[1] = max(1, max(len(s) for s in shards))
        head[2] = max(1, max(sum(len(pairs[i][0]) + len(pairs[i][1]) for i in s) for s in shards))
    name, head = name.to(dev), head.to(dev)
    dist.broadcast(name, src=src)
    dist.broadcast(head, src=src)
    model_type = bytes(name.cpu().numpy().tobytes()).split(b"\0", 1)[0].decode()
    n, n_slots, n_bytes = (int(x) for x in head.cpu().tolist())
    # pair shards: one metadata tensor and one residue tensor per rank
    meta = torch.zeros((n_slots, 3), dtype=torch.int64, device=dev)
    res = torch.zeros(n_bytes, dtype=torch.uint8, device=dev)
    if rank == src:
        packed = [pack_shard(s, pairs, n_slots, n_bytes) for s in shards]
        dist.scatter(meta, [p[0].to(dev) for p in packed], src=src)
        dist.scatter(res, [p[1].to(dev) for p in packed], src=src)
    else:
        dist.scatter(meta, None, src=src)
        dist.scatter(res, None, src=src)
    local = unpack_shard(meta, res)
    results = align(model_type, [p for _, p in local])
    # result gather (every rank gets the list: whichever rank prints does so in submission order)
    gathered = all_gather_ragged(encode_results([(i, r) for (i, _), r in zip(local, results)]), dev)
    out = [None] * n
    for shard in gathered:
        for i, r in decode_results(shard):
            out[i] = r
    return out
