"""bench.py's multi-rank control flow on CPU: `python bench.py --gpus 2` starts its own ranks, they rendezvous
(gloo here, RCCL on the GPUs), time K steps between barriers, reduce the MAX over ranks and rank 0 prints ONE JSON
line.  The engine is a stub (C4_BENCH_STUB=1: no device, no alignment) — this checks the plumbing the driver's
8-GPU run depends on, not a number."""
import json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra, timeout=300):
    env = dict(os.environ, C4_BENCH_STUB="1", **env_extra)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        if k not in env_extra:
            env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, timeout=timeout)


def test_gpus_flag_starts_one_rank_per_gpu_and_prints_one_line():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--pairs", "2"], {})
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.strip()]
    # exactly one JSON line, and it is the last thing on stdout (the comm library may print a banner before it)
    assert len([l for l in lines if l.startswith("{")]) == 1 and lines[-1].startswith("{"), lines
    out = json.loads(lines[-1])
    assert (out["n_gpus"], out["steps"], out["warmup"], out["scaling"]) == (2, 3, 1, "weak")
    # weak scaling: whole-job cells = 2 ranks x 2 pairs x 1001 x 100001 per step
    assert abs(out["value"] * out["ms_per_step"] * 1e-3 - 2 * 2 * 1001 * 100001) < 1e-3 * 2 * 2 * 1001 * 100001
    assert out["data"].startswith("stub") and "cpu_baseline" not in out
    # every timed step stages a fresh batch behind the alignment of the one before: staging is inside `value`
    assert out["staging"]["inside_value"] is True and out["staging_ms"] > 0 and out["staging"]["distinct_batches"] == 4
    assert out["config"]["pairs_per_gpu"] == 2 and out["config"]["pairs_per_step"] == 4
    assert len(out["ranks"]["stage_wait_ms_per_step"]) == 2
    # the timed step is a work-queue step: header broadcast, work-item scatter, result gather of every rank's stream
    wq = out["work_queue"]
    assert "gloo" in wq["collectives"] and wq["alignments_delivered_per_step"] == 2 * 2
    assert wq["result_ints_per_step"] == 2 * (2 * 7 + 2 * 4)


def test_single_rank_and_launcher_environment():
    r = _run(["--steps", "1", "--warmup", "0", "--pairs", "1"], {})
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert json.loads(r.stdout.decode().strip().splitlines()[-1])["n_gpus"] == 1
    # a launcher that disagrees with --gpus is an error, not a silent 1-GPU run
    r = _run(["--gpus", "4", "--steps", "1", "--pairs", "1"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and b"--gpus 4" in r.stderr


def test_strong_scaling_cuts_the_step_into_one_shard_per_rank():
    """--scaling strong (BASELINE config 4: 4 096 cDNAs -> 512 per GPU at 8): --pairs is the whole job's step, rank r aligns
    pairs [r * n / N, (r + 1) * n / N) of every batch; the line says which scaling it ran."""
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--pairs", "6", "--scaling", "strong"], {})
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    out = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1])
    assert out["scaling"] == "strong" and out["n_gpus"] == 2
    assert out["config"]["pairs_per_gpu"] == 3 and out["config"]["pairs_per_step"] == 6
    # whole-job cells per step = 6 pairs, whatever the number of ranks
    assert abs(out["value"] * out["ms_per_step"] * 1e-3 - 6 * 1001 * 100001) < 1e-3 * 6 * 1001 * 100001
    assert out["work_queue"]["alignments_delivered_per_step"] == 6
    # a step that does not divide into the ranks is refused
    r = _run(["--gpus", "2", "--steps", "1", "--pairs", "5", "--scaling", "strong"], {})
    assert r.returncode != 0
