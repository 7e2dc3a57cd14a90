"""CPU: `python bench.py --gpus N` (the driver's command shape) becomes a launcher of N fresh rank processes -- before it makes any
GPU call -- instead of exiting.  The launcher itself is replaced by tests/stub_launcher.py, which echoes its command line."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
STUB = os.path.join(ROOT, "tests", "stub_launcher.py")


def _run(args, **env_extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra)
    return subprocess.run([sys.executable, BENCH] + args, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)


def test_gpus_n_spawns_torch_distributed_run_with_the_same_arguments():
    r = _run(["--gpus", "2", "--steps", "7", "--warmup", "3"], RCV_BENCH_LAUNCHER="%s %s" % (sys.executable, STUB),
             RCV_BENCH_ASSUME_DEVICES="2")
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                      # ONE JSON line on stdout; the launcher's chatter went to stderr
    assert "launcher chatter" in r.stderr
    got = json.loads(lines[0])
    argv = got["stub_argv"]
    assert argv[0] == "--nnodes=1"
    assert argv[argv.index("--nproc-per-node") + 1] == "2"
    assert argv[argv.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 <= int(argv[argv.index("--master-port") + 1]) <= 65535
    k = argv.index(BENCH)
    assert argv[k + 1:] == ["--gpus", "2", "--steps", "7", "--warmup", "3"]
    assert got["ipc"] == "0"                              # dmabuf IPC mode is in the ranks' environment


def test_exit_code_of_the_ranks_is_relayed():
    r = _run(["--gpus", "4"], RCV_BENCH_LAUNCHER="%s %s" % (sys.executable, STUB), RCV_BENCH_ASSUME_DEVICES="8", STUB_EXIT="3")
    assert r.returncode == 3


def test_fewer_devices_than_ranks_is_a_one_line_refusal():
    r = _run(["--gpus", "8"], RCV_BENCH_LAUNCHER="%s %s" % (sys.executable, STUB), RCV_BENCH_ASSUME_DEVICES="1")
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "--gpus 8" in r.stderr and "1 HIP device" in r.stderr


def test_under_a_launcher_world_size_must_match():
    r = _run(["--gpus", "4"], WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr
