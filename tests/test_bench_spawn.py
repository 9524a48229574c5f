"""`python bench.py --gpus N` must start N ranks itself: the parent is a pure
launcher, the children get RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* and own the reference's
contiguous blocks of baselines (run-hydra-pspec.py:268-287).  Runs on CPU through --dry-run, which
goes through the same rank plumbing (gloo group, barrier, max-over-ranks reduction) without a GPU."""
import json
import os
import subprocess
import sys

from conftest import REPO


def _run(extra_args, extra_env=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(extra_env or {})
    return subprocess.run([sys.executable, str(REPO / "bench.py")] + extra_args, env=env, cwd=str(REPO),
                          capture_output=True, text=True, timeout=timeout)


def test_gpus_2_spawns_two_ranks_with_the_reference_blocks():
    r = _run(["--gpus", "2", "--dry-run"], {"HPX_BENCH_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["baselines_total"] == 2048
    assert res["blocks"] == [[0, 1024], [1024, 2048]]
    assert len(set(res["pids"])) == 2
    assert res["max_time"] == 0.002          # max over ranks of (0.001, 0.002)


def test_uneven_split_follows_the_quot_rem_rule():
    r = _run(["--gpus", "3", "--dry-run", "--nbl", "5"])
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert res["blocks"] == [[0, 5], [5, 10], [10, 15]] and res["n_gpus"] == 3


def test_a_failing_rank_fails_the_launcher():
    r = _run(["--gpus", "2", "--dry-run"], {"HPX_BENCH_DRYRUN_FAIL_RANK": "1", "HPX_BENCH_SPAWN_TIMEOUT": "120"})
    assert r.returncode != 0
    assert "ranks failed" in r.stderr


def test_world_size_must_match_gpus():
    r = _run(["--gpus", "2", "--dry-run"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


def test_launcher_never_touches_the_gpu_stack():
    """The launcher branch runs before torch / libhpx are imported (a process that has initialised HIP
    must not start GPU children)."""
    src = (REPO / "bench.py").read_text()
    main = src[src.index("def main():"):]
    spawn_at = main.index("spawn_ranks(args")
    assert "import torch" not in main[:spawn_at] and "hpx" not in main[:spawn_at]
    body = src[src.index("def spawn_ranks"):src.index("def init_ranks")]
    assert "import torch" not in body and "hydra_pspec_amd" not in body and "hpx." not in body


def test_under_torch_distributed_run_the_ranks_are_used_as_given():
    """The driver's launch line (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`):
    WORLD_SIZE is set, so bench.py does not spawn again and the ranks take the same blocks."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(REPO / "bench.py"),
                        "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"],
                       env=env, cwd=str(REPO), capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert res["n_gpus"] == 2 and res["blocks"] == [[0, 1024], [1024, 2048]] and len(set(res["pids"])) == 2


def test_committed_dry_run_records_parse():
    """Every profiles/r*_dry_run_8.json is the launcher's own record of an 8-rank dry run: eight distinct processes, the
    reference's contiguous blocks over 8192 baselines (BASELINE.json config C4).  (An empty file was committed once.)"""
    files = sorted((REPO / "profiles").glob("r*_dry_run_8.json"))
    assert files, "no dry-run record under profiles/"
    for f in files:
        res = json.loads(f.read_text())
        assert res["dry_run"] is True and res["n_gpus"] == 8 and res["baselines_total"] == 8192, f
        assert res["blocks"] == [[1024 * r, 1024 * (r + 1)] for r in range(8)] and len(set(res["pids"])) == 8, f
