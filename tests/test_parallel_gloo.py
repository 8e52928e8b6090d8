"""World-size-2 test of the shard-by-query plumbing on CPU (gloo).  The compute stand-in is the oracle;
on the GPUs the same code runs with the HIP engine and backend "nccl" (= RCCL)."""
import os, socket, sys
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from exonerate_amd import _abi, parallel
    import oracle_lib
    from golden_util import load_set
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = _abi.load()
    params = _abi.Params()
    lib.c4gpu_params_default(params)

    def align(model_type, pairs):
        m = _abi.Model()
        assert lib.c4gpu_model_get(model_type.encode(), 0, 0, params, m) == 0
        return [oracle_lib.find_path(m, params, a, b) for a, b in pairs]

    recs = load_set("est2genome")[:12]
    pairs = [(r["query"], r["target"]) for r in recs] if rank == 0 else []
    out = parallel.distributed_find_path(align, "est2genome", pairs)
    if rank == 1:      # every rank gets the full list (score, region, operations as tensors), in submission order
        import exonerate_amd as ex
        model = ex.Model("est2genome")
        lines = [ex.Alignment.from_parts(model, o["score"], o["region"], o["ops"], len(r["query"]), len(r["target"])).vulgar()
                 for o, r in zip(out, recs)]
        q.put([l.split(" ", 2)[2] for l in lines])
    dist.barrier()
    dist.destroy_process_group()


def test_shard_scatter_gather_preserves_order_and_results():
    from golden_util import load_set
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    recs = load_set("est2genome")[:12]
    assert got == [r["vulgar"].split(" ", 2)[2] for r in recs]


def test_shard_helpers():
    from exonerate_amd import parallel
    assert parallel.shard_bounds(10, 4) == [0, 3, 6, 8, 10]
    shards = parallel.shard_by_cost([5, 1, 1, 1, 4, 4], 2)
    assert sorted(sum(shards, [])) == list(range(6))
    loads = [sum([5, 1, 1, 1, 4, 4][i] for i in s) for s in shards]
    assert abs(loads[0] - loads[1]) <= 1
