"""`python bench.py --gpus N` must start its own ranks (the driver's scaling run calls it exactly so).

CPU rehearsal of the launch path: two gloo ranks, tiny problem, the GPU-only log-likelihood call replaced inside the
child processes by the test stand-in of tests/standin/sitecustomize.py (the product itself has no CPU engine).  What is
checked is the plumbing: the parent spawns the ranks before touching any device, relays rank 0's ONE JSON line and the
exit code, the ranks shard the latents and agree on the step."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, *argv, timeout=600):
    env = dict(os.environ, PLMC_DIST_BACKEND="gloo", PLMC_BENCH_DEVICE="cpu", PLMC_TEST_STANDIN="1", OMP_NUM_THREADS="2")
    env["PYTHONPATH"] = os.path.join(ROOT, "tests", "standin") + os.pathsep + env.get("PYTHONPATH", "")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), env=env, cwd=ROOT, capture_output=True,
                          text=True, timeout=timeout)


def test_bench_gpus2_launches_its_own_ranks():
    r = _run({}, "--gpus", "2", "--n", "256", "--steps", "1", "--warmup", "1", "--no-prof", "--no-cpu-baseline", "--no-options")
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["distributed"]["world_size"] == 2 and res["distributed"]["backend"] == "gloo"
    assert res["steps"] == 1 and res["warmup"] == 1 and res["value"] > 0 and res["scaling"] == "strong"
    assert res["config"]["parallelism"] == "latent-shard x2"


def test_bench_rejects_a_mismatched_world():
    r = _run({"WORLD_SIZE": "3", "RANK": "0", "LOCAL_RANK": "0"}, "--gpus", "2", "--n", "256", "--steps", "1", "--warmup", "0",
             "--no-prof", "--no-cpu-baseline", "--no-options")
    assert r.returncode != 0 and "WORLD_SIZE=3" in (r.stderr + r.stdout)
