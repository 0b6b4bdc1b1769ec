"""Launcher of `python bench.py --gpus N` (open_o3_video_amd/launch.py) on CPU: N child ranks over gloo, rank 0 prints
the world size, a failing or missing rank fails the launch."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=240):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, env=env,
                          timeout=timeout, cwd=ROOT)


def test_bench_self_launches_n_ranks_over_gloo():
    r = _run(["--gpus", "2", "--launch-check"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout          # only rank 0 prints
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["launch_check"] is True


def test_bench_single_rank_check_needs_no_launcher():
    r = _run(["--gpus", "1", "--launch-check"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_world_size_mismatch_is_an_error():
    # under a launcher that set WORLD_SIZE=3 while --gpus says 2, no line may be printed
    r = _run(["--gpus", "2", "--no-cpu-baseline"], env_extra={"WORLD_SIZE": "3", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=3" in (r.stderr + r.stdout)
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_failing_rank_fails_the_launch(tmp_path):
    from open_o3_video_amd.launch import spawn_ranks
    script = tmp_path / "w.py"
    script.write_text("import os, sys, time\nr = int(os.environ['RANK'])\nassert os.environ['WORLD_SIZE'] == '3'\n"
                      "assert os.environ['MASTER_ADDR'] == '127.0.0.1'\nif r == 1:\n    sys.exit(7)\ntime.sleep(30)\n")
    rc = spawn_ranks(3, [str(script)], timeout_s=60)
    assert rc == 7
