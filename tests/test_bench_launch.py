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
    # the N > 1 line's `collective` block: world size as the process group sees it, one entry per rank, every gradient message timed
    # with its ring bus bandwidth (the same code path an RCCL run takes, here over gloo with host tensors)
    c = res["collective"]
    assert c["world_size"] == 2 and sorted(r["rank"] for r in c["ranks"]) == [0, 1] and len({r["pid"] for r in c["ranks"]}) == 2
    assert len(c["messages"]) == 2 and all(m["ms"] > 0 and m["bus_gbps"] > 0 and m["bytes"] > 0 for m in c["messages"])


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


def _retry_env(spec, **kw):
    return dict(_env(), SH_BENCH_TEST_RANK_FAIL=spec, SH_BENCH_FAIL_GRACE="1", SH_BENCH_ATTEMPT_TIMEOUT="120", **kw)


def test_failed_attempt_is_retried_with_fresh_children():
    """A rank that dies in the first attempt must not end the job: every rank's
    GPU-free supervisor starts a FRESH child for the next, more conservative mode on a fresh rendezvous port; rank 0 prints
    exactly one JSON line - the successful attempt's - and says in config.launch what happened before."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0"],
                       env=_retry_env("0:1"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    res = _one_json_line(r.stdout)
    assert res["n_gpus"] == 2
    assert "attempt 1 'eager-safe'" in res["config"]["launch"] and "eager: " in res["config"]["launch"], res["config"]["launch"]
    assert "attempt 0 (eager) failed" in r.stderr


def test_retry_under_torch_distributed_run_rank0_failure():
    """The same under the driver's launcher, with rank 0 (which owns the JSON line) the one that dies: the next mode
    (eager) produces the line, torch.distributed.run sees exit status 0 from every rank."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = _retry_env("0:0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
                        "--warmup", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    res = _one_json_line(r.stdout)
    assert "attempt 1 'eager-safe'" in res["config"]["launch"]


def test_hung_attempt_is_cut_short_by_the_progress_watchdog():
    """A rank that HANGS after the rendezvous (a collective that never completes - the failure mode a first multi-GPU run of
    the hipGraph-with-RCCL attempt could have) is not waited for until the whole-attempt limit (which has to cover a cold import
    on a fresh box): every phase after the rendezvous is seconds of work, so its supervisor gives up after
    SH_BENCH_PHASE_TIMEOUT without progress and the job goes on with the next mode."""
    import time
    env = dict(_env(), SH_BENCH_TEST_RANK_HANG="0:1", SH_BENCH_PHASE_TIMEOUT="3", SH_BENCH_FAIL_GRACE="1", SH_BENCH_ATTEMPT_TIMEOUT="600",
               SH_BENCH_GRAPH_ATTEMPT_TIMEOUT="600")
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert time.time() - t0 < 200
    res = _one_json_line(r.stdout)
    assert "attempt 1 'eager-safe'" in res["config"]["launch"], res["config"]["launch"]
    assert "no progress for 3 s after phase 'rendezvous'" in r.stderr, r.stderr


def test_graph_attempt_is_opt_in_and_then_goes_first():
    """SH_BENCH_DP_GRAPH=1 puts the hipGraph-with-RCCL mode in front of the other two (default since round 5: eager first - the
    graph is no faster in a world of one and is the mode that could hang)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0"],
                       env=_retry_env("0:1", SH_BENCH_DP_GRAPH="1"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    res = _one_json_line(r.stdout)
    assert "attempt 1 'eager'" in res["config"]["launch"] and "graph: " in res["config"]["launch"], res["config"]["launch"]
    assert "attempt 0 (graph) failed" in r.stderr


def test_all_attempts_failing_gives_a_nonzero_status():
    env = _retry_env("0:1")                                  # attempts: eager (index 0, fails), eager-safe (index 1)
    env["SH_BENCH_TEST_RANK_FAIL"] = "*:1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def _tagged_processes(tag):
    """PIDs of live processes whose environment carries SH_BENCH_TEST_TAG=<tag> (the launcher, its supervisors, their ranks)."""
    out = []
    needle = ("SH_BENCH_TEST_TAG=%s" % tag).encode()
    for d in os.listdir("/proc"):
        if not d.isdigit():
            continue
        try:
            with open("/proc/%s/environ" % d, "rb") as f:
                if needle in f.read().split(b"\0"):
                    with open("/proc/%s/stat" % d) as g:
                        if g.read().rsplit(")", 1)[1].split()[0] != "Z":
                            out.append(int(d))
        except OSError:
            continue
    return out


def test_terminated_launcher_leaves_no_rank_behind(tmp_path):
    """a launcher that is terminated (the driver's `timeout`, torch.distributed.run after a rank failure) must not
    orphan rank processes - on a GPU box they would sit in a collective holding the device.  Two ranks hang after the
    rendezvous; the launcher gets SIGTERM; every process of the job (launcher -> supervisors -> ranks) must be gone, and the
    supervisors' marker directory with it."""
    import signal
    import time
    import uuid
    tag = uuid.uuid4().hex
    env = dict(_env(), SH_BENCH_TEST_RANK_HANG="*:*", SH_BENCH_TEST_TAG=tag, TMPDIR=str(tmp_path))
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                         stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    try:
        t0 = time.time()
        while time.time() - t0 < 120 and len(_tagged_processes(tag)) < 5:      # launcher + 2 supervisors + 2 ranks
            time.sleep(0.2)
        assert len(_tagged_processes(tag)) >= 5, _tagged_processes(tag)
        time.sleep(2.0)                                                      # let the ranks reach the rendezvous
        p.send_signal(signal.SIGTERM)
        p.wait(60)
        t0 = time.time()
        while time.time() - t0 < 30 and _tagged_processes(tag):
            time.sleep(0.2)
        assert _tagged_processes(tag) == []
        assert not [d for d in os.listdir(tmp_path) if d.startswith("sh_bench_")], os.listdir(tmp_path)
    finally:
        for pid in _tagged_processes(tag):
            try:
                os.kill(pid, signal.SIGKILL)
            except OSError:
                pass


def test_killed_supervisor_takes_its_rank_along(tmp_path):
    """... and a supervisor that is SIGKILLed (no handler, no finally) still does not leave its rank: the child asked the
    kernel for SIGKILL on its parent's death (PR_SET_PDEATHSIG) before it would touch a GPU."""
    import signal
    import time
    import uuid
    tag = uuid.uuid4().hex
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(_env(), SH_BENCH_TEST_RANK_HANG="*:*", SH_BENCH_TEST_TAG=tag, TMPDIR=str(tmp_path), WORLD_SIZE="2", RANK="1", LOCAL_RANK="1",
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sup = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    try:
        t0 = time.time()
        while time.time() - t0 < 120 and len(_tagged_processes(tag)) < 2:      # the supervisor and its rank
            time.sleep(0.2)
        assert len(_tagged_processes(tag)) == 2
        time.sleep(3.0)                                                      # the rank has set PR_SET_PDEATHSIG by now
        sup.kill()
        sup.wait(30)
        t0 = time.time()
        while time.time() - t0 < 30 and _tagged_processes(tag):
            time.sleep(0.2)
        assert _tagged_processes(tag) == []
    finally:
        for pid in _tagged_processes(tag):
            try:
                os.kill(pid, signal.SIGKILL)
            except OSError:
                pass
