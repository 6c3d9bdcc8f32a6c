"""bench.py's own N > 1 path on CPU: its launcher (`spawn_ranks`, i.e. what `python3 bench.py
--gpus N` does when no torchrun is around it), its whole-ego sharding, its exchange and its JSON,
with two gloo ranks and the oracle standing in for the per-rank HIP compute
(tests/_bench_gloo_entry.py).  SURVEY.md section 8(e): shard + gather == single, bit for bit."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENTRY = os.path.join(REPO, "tests", "_bench_gloo_entry.py")


def _launch(tmp_path, world, extra, env_extra=None, cpu_baseline=False, want_rc=0):
    argv = ["--gpus", str(world), "--steps", "2", "--warmup", "1", "--prewarm-ms", "0", "--no-extra",
            "--horizon", "12", "--dump-gathered", str(tmp_path), *extra]
    if not cpu_baseline:
        argv.append("--no-cpu-baseline")
    code = ("import sys; sys.path.insert(0, %r); import bench; "
            "sys.exit(bench.spawn_ranks(%d, %r, script=%r))" % (REPO, world, argv, ENTRY))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra or {})
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == want_rc, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, f"rank 0 prints ONE JSON line, got {len(lines)}: {res.stdout[-2000:]} {res.stderr[-2000:]}"
    out = json.loads(lines[0])
    out["_stderr"] = res.stderr
    return out


@pytest.mark.parametrize("mode,per_gpu", [("weak", 70), ("strong", 65), ("strong", 70)])
def test_bench_two_ranks_shard_by_whole_egos_and_gather_bitwise(tmp_path, mode, per_gpu, oracle, workloads):
    import bench
    world = 2
    extra = ["--rollouts-per-gpu", str(per_gpu)] + (["--strong"] if mode == "strong" else [])
    out = _launch(tmp_path, world, extra)
    n_total = bench.total_rollouts(world, per_gpu, mode == "strong")
    assert out["n_gpus"] == world and out["world_seen"] == world and out["scaling"] == mode
    assert out["rollouts_total"] == n_total
    shards = out["shards"]
    assert shards == [list(workloads.shard_egos(n_total, world, r)) for r in range(world)]
    assert shards[0][0] == 0 and shards[-1][1] == n_total
    assert all(lo % workloads.NUM_PATHS == 0 for lo, _ in shards), "ranks own whole egos"
    assert out["exchange"]["verified"] is True and out["exchange"]["overlapped"] is True
    assert out["config"]["lanes_per_rollout"] == 1, "shards run the lane-per-rollout kernel (bitwise contract)"
    assert out["value"] > 0 and out["steps"] == 2 and out["unit"] == "vehicle-steps/s"
    if mode == "weak":
        assert n_total == world * (per_gpu // 7) * 7 and len({hi - lo for lo, hi in shards}) == 1
    # the default N > 1 line is complete: both exchanges side by side, the fixed-N split
    # with both kernels, per-GPU rooflines -- all outside the timed region, all verified
    ab = out["exchange_ab"]
    assert ab["steps"] == 1 and ab["rccl"]["available"] and ab["rccl"]["verified"] is True and ab["rccl"]["ms_per_step"] > 0
    assert ab["rccl"]["kind"] == "all_gather_into_tensor"
    assert ab["p2p"]["available"] is False and "handle" in ab["p2p"]["reason"]     # CPU stand-in: no library handle
    st = out["strong"]
    assert st["rollouts_total"] == per_gpu and "error" not in st
    assert st["shards"] == [list(workloads.shard_egos(per_gpu, world, r)) for r in range(world)]
    for k, ln in (("lane", 1), ("wheel_parallel", 4)):
        assert st[k]["verified"] is True and st[k]["ms_per_step"] > 0 and st[k]["lanes_per_rollout"] == ln
    assert out["roofline_hbm"]["frac"] > 0 and out["roofline_hbm"]["per_gpu"] is True and out["roofline"]["per_gpu"] is True
    assert "sections_timed_out" not in out and "_section" not in out
    assert out["exchange_calibration"] is None       # CPU stand-in: no peer copies to compare the collective with
    # what every rank holds after the last exchange == the single-process result, bit for bit
    s0, tab, pid = workloads.config3(n_total, 12, np.float32)
    single = oracle.rollout(oracle.default_params(), s0.astype(np.float64), tab.astype(np.float64), 1e-3,
                            path_id=pid, nthreads=1).astype(np.float32)
    for r in range(world):
        got = np.load(tmp_path / f"gathered_rank{r}.npy")
        assert got.dtype == np.float32 and np.array_equal(got, single)


def test_one_ulp_of_difference_in_a_peer_block_is_reported_as_unverified(tmp_path):
    """`exchange.verified` is more than "my own block came back": every rank integrates the NEXT rank's inputs itself
    and compares the delivered block bitwise.  Rank 1 here returns one element one ulp off; rank 0 must notice, and
    the verdict rank 0 prints is the AND over the ranks."""
    out = _launch(tmp_path, 2, ["--rollouts-per-gpu", "70"], env_extra={"VDYN_TEST_FLIP_RANK": "1"})
    assert out["exchange"]["verified"] is False and "bitwise" in out["exchange"]["verified_how"]


def test_default_multirank_line_carries_the_cpu_baseline(tmp_path):
    """N > 1 without --no-cpu-baseline: rank 0 times the oracle on its host while the other ranks wait at the closing
    barrier; the line carries `cpu_baseline` as at N = 1 (no fp32 state error here: not the full-size workload)."""
    out = _launch(tmp_path, 2, ["--rollouts-per-gpu", "70", "--cpu-baseline-rollouts", "4096"], cpu_baseline=True)
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and "sample" in cb
    assert out["gpu_over_cpu"] > 0 and "fp32_state_error" not in out
    assert out["exchange_ab"]["rccl"]["verified"] is True and out["strong"]["lane"]["verified"] is True


def test_sections_can_be_switched_off(tmp_path):
    out = _launch(tmp_path, 2, ["--rollouts-per-gpu", "70", "--no-sections"])
    assert "exchange_ab" not in out and "strong" not in out and out["exchange"]["verified"] is True


FAKE_P2P = {"VDYN_TEST_FAKE_P2P": "1"}


def test_auto_with_peer_copies_calibrates_and_stays_collective(tmp_path):
    """`auto` with a (stand-in) peer-copy exchange available: the calibration runs, both numbers are maxima over the
    ranks, the choice is the same on every rank (the run would hang otherwise), nothing is relaunched."""
    out = _launch(tmp_path, 2, ["--rollouts-per-gpu", "70", "--run-timeout-s", "200"], env_extra=FAKE_P2P)
    cal = out["exchange_calibration"]
    assert cal["steps"] >= 8 and cal["peer_copies_ms_per_step"] > 0 and cal["all_gather_ms_per_step"] > 0
    assert cal["chosen"] == out["exchange"]["kind"] and "error" not in cal
    assert out["exchange"]["verified"] is True and "relaunched" not in out and out["attempt"] == 1
    assert out["exchange_ab"]["p2p"]["available"] and out["exchange_ab"]["p2p"]["verified"] is True


def test_peer_copy_that_never_completes_ends_in_a_relaunch_with_rccl_and_one_line(tmp_path):
    """the stand-in PeerExchange.start blocks forever on every rank.  Each rank's whole-run
    watchdog (armed before torch is imported) ends it with EXIT_WATCHDOG, rank 0 having printed a `value: null` line
    that names the stage; the supervisors then start FRESH ranks once with --exchange rccl --no-calibration
    --no-peer-copies; ONE line comes out, measured, marked `relaunched`, exit code 0."""
    import bench
    out = _launch(tmp_path, 2, ["--rollouts-per-gpu", "70", "--run-timeout-s", "25"],
                  env_extra=dict(FAKE_P2P, VDYN_TEST_P2P_FAULT="hang_in_start"))
    rl = out["relaunched"]
    assert rl["after"] == bench.EXIT_WATCHDOG and rl["first_attempt_stage"] == "calibration: peer copies"
    # (the rank whose deadline passes first ends, its supervisor marks the attempt failed, the other rank may see that
    # mark a moment before its own deadline)
    assert rl["first_attempt_why"] in ("run_deadline", "peer_rank_failed")
    assert out["attempt"] == 2 and out["value"] > 0 and out["exchange"]["verified"] is True
    assert out["exchange"]["kind"] == "all_gather_into_tensor" and out["exchange"]["peer_copies_disabled"] is True
    assert out["exchange_calibration"] is None
    assert out["exchange_ab"]["p2p"]["available"] is False and "disabled" in out["exchange_ab"]["p2p"]["reason"]
    assert out["strong"]["lane"]["verified"] is True and out["strong"]["lane"]["exchange"] == "all_gather_into_tensor"
    # every rank said on stderr where it was
    assert "watchdog (run_deadline)" in out["_stderr"]
    for r in (0, 1):
        assert f"[bench.py] rank {r}: watchdog (" in out["_stderr"]


@pytest.mark.parametrize("world", [2, 3])
def test_rank_that_raises_after_the_agree_step_ends_in_a_relaunch_without_waiting_out_the_deadline(tmp_path, world):
    """Rank 1 raises in try_create after the agree step (its peers have gone on to the next collective).  It ends
    non-zero; its supervisor marks the attempt failed; rank 0's watchdog sees the mark and ends rank 0 long before its
    200 s deadline; fresh ranks run over RCCL."""
    import time
    t0 = time.time()
    out = _launch(tmp_path, world, ["--rollouts-per-gpu", "70", "--run-timeout-s", "200"],
                  env_extra=dict(FAKE_P2P, VDYN_TEST_P2P_FAULT="raise_after_agree"))
    assert time.time() - t0 < 150, "the surviving ranks must not wait out their deadline"
    assert out["n_gpus"] == world and len(out["shards"]) == world
    rl = out["relaunched"]
    # rank 0 either saw the mark (a rank hung in an RCCL collective would) or its gloo collective raised when rank 1's
    # sockets closed: both leave a line with the stage behind
    assert rl["first_attempt_why"] == "peer_rank_failed" or rl["first_attempt_why"].startswith("exception: ")
    assert rl["first_attempt_stage"] is not None and rl["after"] != 0
    assert out["value"] > 0 and out["attempt"] == 2 and out["exchange"]["kind"] == "all_gather_into_tensor"
    assert out["exchange"]["verified"] is True
    assert "stand-in failure after the agree step" in out["_stderr"]


def test_no_relaunch_when_the_collective_itself_was_asked_for(tmp_path):
    """--exchange rccl and a hang: there is nothing else to fall back to; still ONE line (value null, the stage), and
    a non-zero exit code."""
    import bench
    out = _launch(tmp_path, 2, ["--rollouts-per-gpu", "70", "--run-timeout-s", "20", "--exchange", "rccl"],
                  env_extra={"VDYN_TEST_HANG_STAGE": "headline"}, want_rc=1)
    assert out["value"] is None and out["relaunched"] is None and out["timed_out_in"].startswith("headline")
    assert out["timed_out"]["exit_code"] == bench.EXIT_WATCHDOG


def test_total_rollouts_keeps_one_wave_per_simd():
    import bench
    assert bench.total_rollouts(1, 65536, False) == 65536
    for w in (2, 4, 8):
        n = bench.total_rollouts(w, 65536, False)
        assert n == w * 65534 and n % 7 == 0
        assert bench.total_rollouts(w, 65536, True) == 65536


def test_self_launch_happens_before_torch_is_imported():
    """`python3 bench.py --gpus 2` as typed: the parent must become the launcher without importing
    torch (no HIP initialisation in a process that then starts others)."""
    code = ("import sys, runpy; sys.argv = ['bench.py', '--gpus', '2']\n"
            "import subprocess\n"
            "def fake_run(cmd, env=None):\n"
            "    assert 'torch' not in sys.modules, 'torch imported before the launch'\n"
            "    assert cmd[1:3] == ['-m', 'torch.distributed.run'] and '--nproc-per-node=2' in cmd\n"
            "    assert cmd[-2:] == ['--gpus', '2'] and '127.0.0.1' in cmd\n"
            "    assert env.get('HSA_ENABLE_IPC_MODE_LEGACY') == '0'\n"
            "    class R: returncode = 7\n"
            "    return R()\n"
            "subprocess.run = fake_run\n"
            "runpy.run_path(%r, run_name='__main__')\n" % os.path.join(REPO, "bench.py"))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=env)
    assert res.returncode == 7, res.stderr[-2000:]      # the child's exit code is ours


def test_ending_the_launcher_ends_the_workers_too(tmp_path):
    """The ranks' real work runs in children of the supervisors.  When the job is torn down from above (the agent's
    SIGTERM, or a plain kill of a supervisor: PR_SET_PDEATHSIG), no worker may be left behind holding a GPU."""
    import signal
    import time
    import psutil
    argv = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--prewarm-ms", "0", "--no-extra", "--horizon", "12",
            "--rollouts-per-gpu", "70", "--no-cpu-baseline", "--run-timeout-s", "300"]
    code = ("import sys; sys.path.insert(0, %r); import bench; "
            "sys.exit(bench.spawn_ranks(2, %r, script=%r))" % (REPO, argv, ENTRY))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(FAKE_P2P, VDYN_TEST_P2P_FAULT="hang_in_start")           # the workers sit in their first push for good
    top = subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    try:
        me = psutil.Process(top.pid)
        deadline = time.time() + 120
        workers = []
        while time.time() < deadline:
            procs = me.children(recursive=True)
            workers = [p for p in procs if "VDYN_BENCH_WORKER" in (p.environ() if p.is_running() else {})]
            if len(workers) == 2:
                break
            time.sleep(0.5)
        assert len(workers) == 2, "two worker ranks expected under two supervisors"
        workers.sort(key=lambda w: int(w.environ()["RANK"]))
        supervisors = [w.parent() for w in workers]
        time.sleep(3.0)                                                 # let them reach the hang
        supervisors[1].send_signal(signal.SIGKILL)                      # rank 1's supervisor dies without a word ...
        top.send_signal(signal.SIGTERM)                                 # ... and the launcher is asked to stop
        gone, alive = psutil.wait_procs(workers + supervisors, timeout=60)
        assert not alive, f"left behind: {[(p.pid, p.cmdline()[-3:]) for p in alive]}"
        # rank 0's supervisor, told to go by the agent (SIGTERM), still left ONE line behind: value null, and why
        out, _ = top.communicate(timeout=60)
        lines = [json.loads(ln) for ln in out.splitlines() if ln.startswith("{")]
        assert len(lines) == 1 and lines[0]["value"] is None and lines[0]["terminated_by_signal"] == signal.SIGTERM, out[-2000:]
    finally:
        try:
            for p in psutil.Process(top.pid).children(recursive=True):
                p.kill()
        except psutil.NoSuchProcess:
            pass
        top.kill()
        top.wait(timeout=30)
