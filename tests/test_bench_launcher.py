"""bench.py typed plainly with --gpus N > 1 starts its own ranks: the launcher's command line, the relay of
rank 0's line, and the restart with the blocking gather when the ranks die or hang.  CPU only: nothing here touches a GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_launcher_argv_is_the_drivers_launch_form():
    argv = ["--gpus", "8", "--steps", "5", "--warmup", "2"]
    cmd = bench.launcher_argv(argv, 8, 29511)
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29511"
    script = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[script + 1:] == argv                      # the ranks see the arguments as typed


def test_plain_command_with_n_gt_1_goes_through_the_launcher():
    env = dict(os.environ, RECO_BENCH_DRY_LAUNCH="1")
    env.pop("RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--share-gpus"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    cmd = json.loads(out.stdout.strip().splitlines()[-1])["launcher_argv"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "2" and cmd[-5:] == ["--gpus", "2", "--steps", "1", "--share-gpus"]


def test_more_ranks_than_gpus_is_refused_without_the_rehearsal_flag():
    env = dict(os.environ)
    env.pop("RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True,
                         timeout=300)
    assert out.returncode != 0 and "visible" in out.stderr


def test_result_line_is_the_last_json_object():
    text = "RCCL version 2.x\n{\"metric\": \"m\", \"value\": 1}\ntrailing noise\n"
    assert json.loads(bench._result_line(text))["value"] == 1
    assert bench._result_line("no json here\n{broken\n") is None


def test_failed_or_hung_ranks_are_restarted_with_the_blocking_gather(monkeypatch, capsys):
    calls = []

    def fake_run(cmd, env, timeout_s):
        calls.append(env.get("RECO_BENCH_SYNC_GATHER"))
        if len(calls) == 1:
            return None, "", []                              # first attempt: hung until the limit, everything cleaned up
        return 0, "banner\n" + json.dumps({"metric": "m", "value": 2.0, "n_gpus": 2}) + "\n", []

    monkeypatch.setattr(bench, "_run_children", fake_run)
    monkeypatch.setattr(bench, "visible_gpus", lambda: 2)
    monkeypatch.delenv("RECO_BENCH_SYNC_GATHER", raising=False)
    monkeypatch.delenv("RECO_BENCH_DRY_LAUNCH", raising=False)
    argv = ["--gpus", "2"]
    assert bench.self_launch(bench.parse_args(argv), argv) == 0
    assert calls == [None, "1"]
    out = capsys.readouterr()
    assert json.loads(out.out.strip())["n_gpus"] == 2 and "timed out" in out.err


def test_hung_children_are_killed_by_process_group():
    rc, out, survivors = bench._run_children([sys.executable, "-c", "import time; print('x', flush=True); time.sleep(60)"], dict(os.environ), 1.0)
    assert rc is None and out.strip() == "x" and survivors == []


def test_hung_ranks_in_sessions_of_their_own_are_killed_too(tmp_path):
    """torch elastic starts every rank with start_new_session=True: the ranks are not in the agent's process group.  A fake agent
    that ignores SIGTERM and whose grandchild called setsid() must still leave nothing behind."""
    pidfile = tmp_path / "rank.pid"
    rank = "import os, time; os.setsid(); open(%r, 'w').write(str(os.getpid())); time.sleep(120)" % str(pidfile)
    agent = ("import signal, subprocess, sys, time; signal.signal(signal.SIGTERM, signal.SIG_IGN); "
             "subprocess.Popen([sys.executable, '-c', %r]); print('up', flush=True); time.sleep(120)" % rank)
    rc, out, survivors = bench._run_children([sys.executable, "-c", agent], dict(os.environ), 3.0, grace_s=1.0)
    assert rc is None and out.strip() == "up" and survivors == []
    pid = int(pidfile.read_text())
    assert not bench._alive(pid)


def test_surviving_ranks_stop_the_retry(monkeypatch, capsys):
    calls = []

    def fake_run(cmd, env, timeout_s):
        calls.append(1)
        return None, "", [4242]

    monkeypatch.setattr(bench, "_run_children", fake_run)
    monkeypatch.setattr(bench, "visible_gpus", lambda: 2)
    monkeypatch.delenv("RECO_BENCH_SYNC_GATHER", raising=False)
    monkeypatch.delenv("RECO_BENCH_DRY_LAUNCH", raising=False)
    argv = ["--gpus", "2"]
    assert bench.self_launch(bench.parse_args(argv), argv) == 1
    assert len(calls) == 1 and "survived" in capsys.readouterr().err


def test_visible_gpus_reads_the_environment_before_anything_else(monkeypatch):
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1,2")
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "0,1")
    assert bench.visible_gpus() == 2
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")
    assert bench.visible_gpus() == 0
