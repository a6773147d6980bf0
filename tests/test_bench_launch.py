"""`python bench.py --gpus N` starts its own rank processes when no launcher set WORLD_SIZE (the shape of the
driver's BENCH command).  Here: the launch / rendezvous / timing / reporting skeleton over gloo, no GPU work
(SH_BENCH_DRYRUN), for N = 2, both self-launched and under torch.distributed.run."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = dict(os.environ, SH_BENCH_DRYRUN="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def _one_json_line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_self_launch_two_ranks():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    res = _one_json_line(r.stdout)
    assert res["n_gpus"] == 2 and res["steps"] == 3 and res["config"]["global_batch"] == 128 and res["dry_run"]


def test_under_torch_distributed_run():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
                        "--warmup", "0"], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert _one_json_line(r.stdout)["n_gpus"] == 2


def test_failing_rank_propagates_exit_status():
    env = dict(_env(), WORLD_SIZE="3", RANK="0")          # launcher environment that disagrees with --gpus: exit status 2
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True,
                       timeout=120)
    assert r.returncode == 2
