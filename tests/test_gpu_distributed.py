"""The multi-GPU plumbing on the real backend with the one GPU a test box has: backend "nccl" (= RCCL), world size 1.
exonerate_amd.parallel.distributed_find_path (job header broadcast, pair-shard scatter, result gather — all tensors on the
device) around the HIP engine, against the oracle; and bench.py's work-queue step with a forced process group
(C4_BENCH_FORCE_DIST=1).  The world-size-2 form of the same code runs on gloo in tests/test_parallel_gloo.py and
tests/test_bench_control_flow.py."""
import json, os, socket, subprocess, sys
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import torch, torch.distributed as dist
import exonerate_amd as ex
from exonerate_amd import parallel
from golden_util import load_set
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
eng = ex.Engine(0)
def align(model_type, pairs):
    return eng.find_path(ex.Model(model_type), pairs)
recs = load_set("est2genome")[:12]
pairs = [(r["query"], r["target"]) for r in recs]
out = parallel.distributed_find_path(align, "est2genome", pairs)
model = ex.Model("est2genome")
lines = [ex.Alignment.from_parts(model, o["score"], o["region"], o["ops"], len(r["query"]), len(r["target"])).vulgar().split(" ", 2)[2]
         for o, r in zip(out, recs)]
assert parallel.collective_device().type == "cuda"
dist.barrier()
dist.destroy_process_group()
eng.close()
print("RESULT " + json.dumps(lines))
'''


def _env():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    return dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")


def test_distributed_find_path_on_rccl_with_device_tensors():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from golden_util import load_set
    r = subprocess.run([sys.executable, "-c", WORKER % {"root": ROOT}], env=_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    line = [l for l in r.stdout.decode().splitlines() if l.startswith("RESULT ")][-1]
    recs = load_set("est2genome")[:12]
    assert json.loads(line[7:]) == [x["vulgar"].split(" ", 2)[2] for x in recs]


def test_bench_work_queue_step_on_rccl():
    env = _env()
    env.update(C4_BENCH_FORCE_DIST="1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--pairs", "64", "--tlen", "50000",
                        "--no-revcomp"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    out = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1])
    wq = out["work_queue"]
    assert "RCCL" in wq["collectives"] and wq["alignments_delivered_per_step"] == 64 and wq["probe_pair_identical_on_all_ranks"] is True
    assert out["n_gpus"] == 1 and out["steps"] == 2
    # the one-core reference leg ran before the process group was formed and agrees with the GPU's vulgar line
    assert out["cpu_baseline"]["kind"] == "reference" and out["cpu_baseline"]["vulgar_identical_to_gpu"] is True
