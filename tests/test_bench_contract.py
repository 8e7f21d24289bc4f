"""bench.py's launch contract, on CPU: `--gpus N` started without a launcher spawns N ranks itself (or refuses when fewer than N
GPUs are visible -- never a silent 1-GPU run), and the multi-rank skeleton (rendezvous, barriers, MAX over ranks, ONE JSON line from
rank 0 with n_gpus = N) works over gloo with world_size 2 (`--dry-run`: no device work, marked as not a measurement)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["CUDA_VISIBLE_DEVICES"] = env["HIP_VISIBLE_DEVICES"] = ""
    return env


def test_gpus_n_without_n_gpus_fails_loudly():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, env=_env(), timeout=300)
    assert r.returncode != 0
    assert "refusing to run fewer ranks" in r.stderr and not r.stdout.strip()


def test_gpus_2_dry_run_spawns_two_ranks_and_prints_one_line():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"], capture_output=True, text=True,
                       env=_env(), timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config"):
        assert key in j, key
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["warmup"] == 1 and j["dry_run"] is True and "NOT a measurement" in j["data"]
    assert j["ms_per_step"] >= 2.0  # rank 1 sleeps 2 ms per step: the MAX over ranks, not rank 0's own 1 ms
    assert j["scaling"] == "weak" and "workload" in j["config"]


def test_mismatched_world_size_is_an_error():
    env = _env()
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29512")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_gpus_8_dry_run_through_the_drivers_launcher():
    """The driver's own command for the 8-GPU line -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1
    --master-port P bench.py --gpus 8 ...` -- on CPU (`--dry-run`, gloo): eight ranks rendezvous, ONE line from rank 0 with n_gpus = 8
    and the MAX over ranks (round-5 verdict item 7: the N = 8 path had never executed in any form)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), BENCH, "--gpus", "8", "--steps", "2", "--warmup", "1", "--dry-run"],
                       capture_output=True, text=True, env=_env(), timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and j["steps"] == 2 and j["dry_run"] is True and j["scaling"] == "weak"
    assert j["ms_per_step"] >= 8.0  # rank r sleeps (r + 1) ms per step: rank 7's 8 ms is the step, not rank 0's 1 ms
