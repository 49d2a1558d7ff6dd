"""bench.py prints exactly one JSON line with the fields the driver depends on."""

import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
            "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
            "roofline", "cpu_baseline"}


def run_bench(*extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *extra],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_json_contract():
    d = run_bench("--workload", "c2", "--n", "2000000", "--steps", "2", "--warmup", "1",
                  "--cpu-sample", "200000")
    assert REQUIRED <= set(d)
    assert d["unit"] == "particles/s" and d["higher_is_better"] is True
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["value"] > 0 and abs(d["value"] - 2000000 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0 < r["frac"] < 1
    assert "traffic" in r
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["unit"] == "particles/s" and c["value"] > 0
    assert "sample" in c
    # the same restatement once with one thread and once with all host cores
    s1 = c["single_thread"]
    assert s1["cores"] == 1 and s1["value"] > 0 and "sample" in s1


@pytest.mark.gpu
def test_bench_forced_distributed_path_matches_single():
    """The N>1 code path with one rank produces the same tree and lists."""
    a = run_bench("--workload", "c2", "--n", "3000000", "--steps", "1", "--cpu-sample", "0")
    b = run_bench("--workload", "c2", "--n", "3000000", "--steps", "1", "--cpu-sample", "0",
                  "--force-dist")
    for key in ("nboxes", "nlevels", "list1_entries", "list2_entries"):
        assert a["config"][key] == b["config"][key], key
    assert b["config"]["global_nboxes"] == a["config"]["nboxes"]
