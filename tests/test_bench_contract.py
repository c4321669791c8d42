"""bench.py's output contract (the driver parses it): one JSON line from rank 0 with the agreed keys."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_help_and_workloads_without_a_gpu():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--config"):
        assert flag in r.stdout
    src = open(os.path.join(ROOT, "bench.py")).read()
    for cfg in ("c2", "c3", "c4", "c5"):
        assert f'"{cfg}": dict(' in src


@pytest.mark.gpu
@pytest.mark.parametrize("config", ["c2", "c5"])
def test_bench_line_has_the_contract_keys(config):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", config, "--steps", "3", "--warmup", "1", "--headline-only",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "rank 0 prints exactly ONE line on stdout"
    d = json.loads(lines[0])
    assert d["metric"] == "composed queries/sec" and d["unit"] == "queries/sec"
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["value"] > 0 and abs(d["value"] - 64 * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-6
    assert d["dtype"] == ("f32" if config == "c2" else "fp8")
    assert "workload" in d["config"] and "model" not in d["config"]
    roof = d["roofline"]
    assert roof["bound"] in ("hbm", "mfma") and roof["unit"] in ("GB/s", "TFLOP/s")
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9 and "traffic" in roof


@pytest.mark.gpu
def test_bench_multi_rank_preflight_and_latency_keys():
    """`bench.py --gpus 2` (two ranks sharing this box's one GPU over gloo: the debug layout; on a multi-GPU node the same command is
    one rank per GPU over RCCL): the pre-flight block reports every rank's device and the all-gather rate, two ranks on one device
    are refused without the debug switch, and the line carries the per-batch latency and the roofline regime label."""
    env = dict(os.environ, FERN_DIST_BACKEND="gloo", FERN_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "c4", "--steps", "2", "--warmup", "1", "--headline-only",
           "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "[bench preflight]" in r.stderr
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    pf = d["preflight"]
    assert d["n_gpus"] == 2 and len(pf["ranks"]) == 2 and {e["rank"] for e in pf["ranks"]} == {0, 1}
    assert pf["all_gather_64MiB_per_rank"]["payload_ok"] and pf["all_gather_64MiB_per_rank"]["bytes_received_per_rank"] == 64 << 20
    lat = d["latency_ms_per_batch"]
    assert lat["p50"] > 0 and lat["p99"] >= lat["p50"] and lat["batches"] == 2
    assert d["roofline"]["regime"].startswith("serial_passes") and d["roofline"]["step_level"]["frac"] <= 1.0
    env.pop("FERN_BENCH_SHARE_GPU")
    r2 = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r2.returncode != 0      # one GPU here: rank 1 has no cuda:1 (on a node where two ranks name one device the pre-flight refuses)
