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


def test_roofline_block_fields():
    """The `roofline` object of the bench line from synthetic per-op rows: dominant label, its bound chosen by arithmetic intensity,
    the direct-FLOP pricing note of the Winograd family with the executed rate beside it, and the per-source-kernel `top_families`."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", BENCH)
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    rows = [
        {"label": "conv_wino<64,80>", "kind": 1, "ms": 0.08, "flops": 11.325e9, "bytes": 67.4e6, "shape": "", "bwd": False},
        {"label": "conv_wino<64,80>", "kind": 1, "ms": 0.08, "flops": 11.325e9, "bytes": 67.4e6, "shape": "", "bwd": True},
        {"label": "wgrad_mfma<2,2,2,2,1,f0>", "kind": 3, "ms": 0.05, "flops": 11.325e9, "bytes": 59e6, "shape": "", "bwd": True},
        {"label": "wgrad_mfma<1,1,1,1,4,f0>", "kind": 3, "ms": 0.09, "flops": 11.3e9, "bytes": 471e6, "shape": "", "bwd": True},
        {"label": "tconvms_mfma<2,5,16>", "kind": 2, "ms": 0.10, "flops": 5.6e9, "bytes": 471e6, "shape": "", "bwd": False},
        {"label": "combine", "kind": 9, "ms": 0.05, "flops": 0.0, "bytes": 943e6, "shape": "", "bwd": False},
    ]
    r = bench.roofline_block(rows, 0.5, "robo_unet_640x480_bs32", 1, False)      # (batch override: no PMC file lookup)
    assert r["kernel"] == "conv_wino<64,80>" and r["bound"] == "mfma" and r["launches_per_step"] == 2
    assert abs(r["achieved"] - 2 * 11.325e9 / 0.16e-3 / 1e12) < 0.01 and abs(r["frac"] - r["achieved"] / 157.3) < 1e-3
    assert "direct-conv" in r["flop_pricing"] and "16/36" in r["flop_pricing"]
    assert abs(r["executed_tflops"] - r["achieved"] * 16 / 36) < 0.01 and abs(r["mfma_pipe_frac"] - r["executed_tflops"] / 157.3) < 1e-3
    fam = {f["kernel"]: f for f in r["top_families"]}
    assert set(fam) == {"conv_wino_kernel", "wgrad_mfma_kernel", "convs_mfma_kernel", "combine_kernel"}
    assert fam["wgrad_mfma_kernel"]["launches_per_step"] == 2 and abs(fam["wgrad_mfma_kernel"]["ms_per_step"] - 0.14) < 1e-9
    assert fam["convs_mfma_kernel"]["bound"] == "hbm" and fam["combine_kernel"]["bound"] == "hbm"
    assert abs(sum(f["share"] for f in r["top_families"]) - 1.0) < 1e-3
    # per-layer roofline: sum of max(FLOPs / 157.3 TF/s, bytes / 8 TB/s) over ALL rows
    want = sum(max(x["flops"] / 157.3e9, x["bytes"] / 8000e6) for x in rows)
    assert abs(r["t_roof_ms"] - want) < 1e-3 and abs(r["step_frac"] - want / 0.5) < 1e-3
