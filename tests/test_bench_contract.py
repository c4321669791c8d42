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


CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                 "data", "config", "roofline")


def _canned_full_records():
    """Full records of real runs kept under profiles/ (the round-5 default line the driver could not parse: 24.6 KB, and a c5 line)."""
    out = []
    for name in ("r05_bench_c2_final_tree.json", "r05_bench_c5.json", "r05_bench_c3.json"):
        with open(os.path.join(ROOT, "profiles", name)) as f:
            out.append((name, json.load(f)))
    return out


def test_compact_record_fits_the_cap_and_keeps_the_contract():
    """the ONE stdout line the driver parses must stay small whatever the full record grows to."""
    sys.path.insert(0, ROOT)
    import bench
    assert bench.COMPACT_CAP_BYTES <= 4096
    for name, full in _canned_full_records():
        assert len(json.dumps(full)) > 4096 or name != "r05_bench_c2_final_tree.json"
        c = bench.compact_record(full)
        line = json.dumps(c)
        assert len(line) <= bench.COMPACT_CAP_BYTES, (name, len(line))
        assert "\n" not in line and json.loads(line) == c          # one line, valid JSON, no NaN / Infinity tokens
        for k in CONTRACT_KEYS:
            assert k in c, (name, k)
        assert c["value"] == pytest.approx(full["value"], rel=1e-5) and c["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-5)
        assert "workload" in c["config"] and "model" not in c["config"] and c["config"]["name"] == full["config"]["name"]
        roof = c["roofline"]
        assert set(("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel")) <= set(roof)
        assert roof["frac"] == pytest.approx(roof["achieved"] / roof["peak"], rel=1e-3)
        assert all(len(v) < 130 for v in roof.values() if isinstance(v, str)), "no prose in the compact roofline"
        if full.get("cpu_baseline"):
            cb = c["cpu_baseline"]
            assert set(("value", "unit", "cores", "kind", "sample")) <= set(cb) and cb["kind"] in ("port", "reference")
    # the c2 default record: one number per extra leg survives
    c = bench.compact_record(_canned_full_records()[0][1])
    assert set(c["other_configs"]) == {"c3", "c4", "c5"} and all("value" in v and "roofline_frac" in v for v in c["other_configs"].values())
    assert c["harness"]["generate_fiq_val_predictions_qps"] > 0 and c["pcie_inclusive"]["value"] > 0 and "dR50_pp" in c["modes"]["mx8"]


def test_compact_record_drops_extras_before_contract_keys():
    sys.path.insert(0, ROOT)
    import bench
    full = dict(_canned_full_records()[0][1])
    full["other_configs"] = {f"c{i}": dict(full["other_configs"]["c5"]) for i in range(3, 60)}      # a record that cannot fit
    c = bench.compact_record(full)
    assert len(json.dumps(c)) <= bench.COMPACT_CAP_BYTES
    for k in CONTRACT_KEYS + ("cpu_baseline",):
        assert k in c
    # non-finite numbers never reach the line
    full2 = dict(_canned_full_records()[0][1])
    full2["roofline"] = dict(full2["roofline"], traffic=None, gemm_ms_per_step=float("nan"))
    assert "NaN" not in json.dumps(bench.compact_record(full2))


def _run_bench(tmp_path, *flags, env=None):
    full_path = str(tmp_path / "full.json")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags, "--full-record", full_path], capture_output=True, text=True,
                       timeout=900, cwd=ROOT, env=env)
    return r, full_path


@pytest.mark.gpu
@pytest.mark.parametrize("config", ["c2", "c5"])
def test_bench_line_has_the_contract_keys(config, tmp_path):
    r, full_path = _run_bench(tmp_path, "--config", config, "--steps", "3", "--warmup", "1", "--headline-only", "--no-cpu-baseline")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "rank 0 prints exactly ONE line on stdout"
    assert len(lines[0]) <= 4096
    d = json.loads(lines[0])
    full = json.load(open(full_path))
    assert full["value"] == pytest.approx(d["value"], rel=1e-5) and "regime" in full["roofline"] and "gemm_tiles" in full
    assert d["metric"] == "composed queries/sec" and d["unit"] == "queries/sec"
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["value"] > 0 and abs(d["value"] - 64 * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-4      # the compact line rounds to 6 digits
    assert d["dtype"] == ("f32" if config == "c2" else "fp8")
    assert "workload" in d["config"] and "model" not in d["config"]
    roof = d["roofline"]
    assert roof["bound"] in ("hbm", "mfma") and roof["unit"] in ("GB/s", "TFLOP/s")
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3 and "traffic" in roof


@pytest.mark.gpu
@pytest.mark.slow
def test_bench_default_command_prints_one_compact_line(tmp_path):
    """The command the DRIVER runs (no flags but the step counts): cpu_baseline, the secondary legs and the c3 / c4 / c5 child runs
    all happen, and stdout still ends in ONE line under the cap that carries `roofline` + `cpu_baseline`."""
    import time
    t0 = time.time()
    r, full_path = _run_bench(tmp_path, "--steps", "3", "--warmup", "1")
    wall = time.time() - t0
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and len(lines[0]) <= 4096
    d = json.loads(lines[0])
    for k in CONTRACT_KEYS + ("cpu_baseline",):
        assert k in d
    assert d["steps"] == 3 and d["warmup"] == 1 and d["n_gpus"] == 1 and d["config"]["name"] == "c2"
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1 and d["cpu_baseline"]["value"] > 0
    assert d["cpu_baseline"]["parity_vs_hip"]["rows_with_identical_order"] == d["cpu_baseline"]["parity_vs_hip"]["queries"]
    assert set(d["other_configs"]) == {"c3", "c4", "c5"} and all("error" not in v for v in d["other_configs"].values())
    full = json.load(open(full_path))
    assert "all_schedulable_cores" in full["cpu_baseline"] and full["cpu_baseline"]["all_schedulable_cores"] is None      # behind --cpu-all-cores now
    assert "[bench full record] " in r.stderr
    assert wall < 150, f"default bench took {wall:.0f} s"


@pytest.mark.gpu
def test_bench_multi_rank_preflight_and_latency_keys(tmp_path):
    """`bench.py --gpus 2` (two ranks sharing this box's one GPU over gloo: the debug layout; on a multi-GPU node the same command is
    one rank per GPU over RCCL): the pre-flight block reports every rank's device and the all-gather rate, two ranks on one device
    are refused without the debug switch, and the line carries the per-batch latency and the roofline regime label."""
    env = dict(os.environ, FERN_DIST_BACKEND="gloo", FERN_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    full_path = str(tmp_path / "full2.json")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "c4", "--steps", "2", "--warmup", "1", "--headline-only",
           "--no-cpu-baseline", "--full-record", full_path]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "[bench preflight]" in r.stderr
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    assert len(line) <= 4096 and json.loads(line)["n_gpus"] == 2 and "all_gather" in json.loads(line)
    d = json.load(open(full_path))
    pf = d["preflight"]
    assert d["n_gpus"] == 2 and len(pf["ranks"]) == 2 and {e["rank"] for e in pf["ranks"]} == {0, 1}
    assert pf["all_gather_64MiB_per_rank"]["payload_ok"] and pf["all_gather_64MiB_per_rank"]["bytes_received_per_rank"] == 64 << 20
    lat = d["latency_ms_per_batch"]
    assert lat["p50"] > 0 and lat["p99"] >= lat["p50"] and lat["batches"] == 2
    assert d["roofline"]["regime"].startswith("serial_passes") and d["roofline"]["step_level"]["frac"] <= 1.0
    env.pop("FERN_BENCH_SHARE_GPU")
    r2 = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r2.returncode != 0      # one GPU here: rank 1 has no cuda:1 (on a node where two ranks name one device the pre-flight refuses)


def test_mixed_peak_prices_each_part_of_a_paired_launch_against_its_own_peak():
    """bench.mixed_peak: the c5 default mode's block-scaled launches carry the text tower's bf16 GEMMs (fern_prof_stats.gemm_mx8_bf16_flops);
    the roofline peak of such launches is total flops / (fp8 flops / 5 PFLOP/s + bf16 flops / 2.5 PFLOP/s)."""
    import bench
    assert bench.mixed_peak({"gemm_mx8_flops": 0.0}) == bench.MX8_MFMA_PEAK_TFLOPS
    assert bench.mixed_peak({"gemm_mx8_flops": 3e12, "gemm_mx8_bf16_flops": 0.0}) == bench.MX8_MFMA_PEAK_TFLOPS
    assert abs(bench.mixed_peak({"gemm_mx8_flops": 2e12, "gemm_mx8_bf16_flops": 2e12}) - bench.BF16_MFMA_PEAK_TFLOPS) < 1e-9
    p = bench.mixed_peak({"gemm_mx8_flops": 2.335e12, "gemm_mx8_bf16_flops": 0.372e12})      # a c5 step: 1.963 TFLOP fp8 + 0.372 TFLOP bf16
    assert abs(p - 2.335 / (1.963 / 5000.0 + 0.372 / 2500.0)) < 1e-6 and 4300 < p < 4330
