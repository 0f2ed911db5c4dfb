"""bench.py's own launcher (CPU box, gloo): `python bench.py --gpus 2` with no WORLD_SIZE in the environment starts its two
ranks itself, they rendezvous on 127.0.0.1 and all-reduce; a failing rank makes the parent exit non-zero instead of hanging."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env=None, args=("--gpus", "2", "--launch-check")):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(HB_DIST_BACKEND="gloo", **(extra_env or {}))
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, capture_output=True, text=True,
                          timeout=300)


def test_bench_starts_its_own_ranks():
    r = _run()
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout            # exactly one JSON line, from rank 0
    out = json.loads(lines[0])
    assert out == {"launch_check": True, "world": 2, "backend": "gloo", "sum": 3.0}


def test_bench_failed_rank_gives_nonzero_exit():
    r = _run({"HB_BENCH_FAIL_RANK": "1"})
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_bench_under_a_launcher_keeps_its_world():
    """Started with WORLD_SIZE already set (torch.distributed.run does that), bench.py must NOT spawn again."""
    env = {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0", "MASTER_PORT": "29999"}
    r = _run(env, args=("--gpus", "1", "--launch-check"))
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])["world"] == 1


def test_bench_options_and_hardware_queue_default():
    """The driver's flags and this round's additions parse; importing bench.py sets GPU_MAX_HW_QUEUES=2 unless the environment
    already chose a value (with ROCm's default of 4 the multi-rank update path ran 3x slower); without a GPU the
    benchmark itself refuses to run instead of falling back to anything."""
    code = ("import os, sys; sys.argv=['bench.py','--gpus','1','--steps','20','--warmup','5','--actor-lag','1','--event-every','4',"
            "'--no-async-variant','--vanilla']; import bench; a=bench.parse(); "
            "print(os.environ['GPU_MAX_HW_QUEUES'], a.steps, a.warmup, a.actor_lag, a.event_every, a.prime, a.vanilla)")
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.split() == ["2", "20", "5", "1", "4", "24", "True"]
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=dict(env, GPU_MAX_HW_QUEUES="4"), capture_output=True, text=True,
                       timeout=300)
    assert r.stdout.split()[0] == "4"
    import torch

    if not torch.cuda.is_available():
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1"], env=env,
                           capture_output=True, text=True, timeout=300)
        assert r.returncode != 0 and "no CPU fallback" in (r.stderr + r.stdout)


import pytest  # noqa: E402


@pytest.mark.gpu
def test_bench_line_keeps_the_measurement_contract():
    """`python bench.py` on the GPU: exactly one JSON line on stdout with the contract's keys (metric / value / unit / n_gpus /
    steps / warmup / ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config.workload), the roofline
    object of the dominant kernel (bound, achieved, peak, unit, frac, traffic) and the CPU baselines; value = games x steps / time."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2", "--games", "2048",
                        "--cpu-sample-games", "256", "--cpu-sample-steps", "20"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "cpu_baseline_learner", "grad_steps_per_sec"):
        assert k in d, k
    assert d["metric"] == "env_steps_per_sec" and d["unit"] == "env-steps/s" and d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 2048 * 6 / (d["ms_per_step"] * 6 / 1e3)) <= 1e-6 * d["value"]
    r_ = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r_, k
    assert r_["bound"] == "hbm" and r_["unit"] == "GB/s" and abs(r_["frac"] - r_["achieved"] / r_["peak"]) < 1e-9
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    assert d["async_actor"]["actor_lag"] == 1 and d["async_actor"]["env_steps_per_sec"] > 0
