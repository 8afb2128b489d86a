"""bench.py started plainly with --gpus N must measure N GPUs: it starts the ranks itself, as a child
process of torch.distributed.run, and relays rank 0's line.  These tests cover the command it builds and
the refusal to print a one-GPU line under an N-GPU request; the ranks themselves need GPUs."""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_rank_launch_command():
    bench = _bench()
    argv = ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    cmd = bench.rank_launch_command(8, argv, 29511)
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29511"
    script = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[script + 1:] == argv  # the ranks get the caller's own arguments, --gpus included


def test_world_size_must_match_gpus():
    """A rank environment of 2 with --gpus 4 is refused before anything is measured (no GPU is touched: the check
    comes first for a world of one; here the process group cannot even form, so run with WORLD_SIZE=1)."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--no-cpu"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0
    assert "--gpus 4 but the launcher started 1 rank" in (p.stderr + p.stdout)
    assert '"metric"' not in p.stdout


def test_lavapipe_probe_reports_a_verdict():
    verdict = _bench().lavapipe_probe()
    assert verdict["lavapipe"] == "unavailable" or verdict["lavapipe"].startswith("present")
    assert set(verdict["lavapipe_found"]) == {"loader", "icd"}
