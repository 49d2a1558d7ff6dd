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


def run_bench(*extra, launcher=()):
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_PORT")}
    out = subprocess.run([sys.executable, *launcher, os.path.join(ROOT, "bench.py"), *extra],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
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


@pytest.mark.gpu
def test_bench_forced_distributed_path_with_target_extents():
    """BASELINE configs[3] (targets with radii) through the N > 1 code path on one rank: the
    library's multi-GPU entries carry the radii, and the tree and lists are the single-GPU ones."""
    a = run_bench("--workload", "c4", "--n", "2000000", "--steps", "1", "--cpu-sample", "0")
    b = run_bench("--workload", "c4", "--n", "2000000", "--steps", "1", "--cpu-sample", "0",
                  "--force-dist")
    for key in ("nboxes", "nlevels", "list1_entries", "list2_entries"):
        assert a["config"][key] == b["config"][key], key
    assert b["config"]["global_nboxes"] == a["config"]["nboxes"]
    assert "bt_mgpu" in b["config"]["sharded_traversal"]


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def check_two_ranks(d, n):
    assert d["n_gpus"] == 2
    c = d["config"]
    assert c["workload"].startswith("c5:")            # BASELINE configs[4] is the N > 1 default
    assert c["collectives"]["ranks_in_group"] == 2
    assert len(c["owned_particles_by_rank"]) == 2 and sum(c["owned_particles_by_rank"]) == 2 * n
    assert 1.0 <= c["owned_particle_imbalance"] < 1.2
    assert c["exchange_bytes"] == sum(c["exchange_bytes_by_rank"]) > 0
    # uniform points, two owners: about half of a rank's particles leave it -- 24 bytes of
    # coordinates and, in a message of its own, the 4-byte global user id
    assert 0.3 * 28 * n < c["exchange_bytes_by_rank"][0] < 0.7 * 28 * n
    assert c["exchange_bytes"] == c["exchange_bytes_coordinates"] + c["exchange_bytes_ids"]
    assert c["exchange_bytes_ids"] * 6 == c["exchange_bytes_coordinates"]


def test_bench_gpus_2_starts_two_ranks_cpu():
    """`bench.py --gpus 2` without a launcher starts two ranks itself (gloo on the CPU here;
    --dry-run: rendezvous + particle exchange, the hot path has no CPU fallback)."""
    d = run_bench("--gpus", "2", "--dry-run", "--n", "20000")
    assert d["dry_run"] is True and d["value"] is None
    check_two_ranks(d, 20000)


def test_bench_under_torch_distributed_run_cpu():
    """The driver's N > 1 command line: one rank per process from torch.distributed.run."""
    d = run_bench("--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run",
                  "--points-per-gpu", "20000",
                  launcher=("-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                            "--master-addr", "127.0.0.1", "--master-port", str(_free_port())))
    check_two_ranks(d, 20000)


def test_bench_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a machine without a HIP device")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--n", "1000"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode != 0 and "no CPU fallback" in out.stderr


@pytest.mark.gpu
def test_bench_gpus_2_real_processes_one_gpu():
    """Two REAL rank processes through the whole N > 1 path -- exchange, per-rank build,
    global numbering, local essential tree, lists -- on however many GPUs the box has (one:
    the ranks share it -- RCCL refuses two ranks on a device -- and drive the library's bt_mgpu_*
    entries over its shared-memory communicator; torch.distributed / gloo only carries the
    barrier and the statistics).  The global tree of the two shards must be the tree one rank
    builds from both chunks, and the job must say that it timed the library's implementation."""
    n = 400000
    d = run_bench("--gpus", "2", "--n", str(n), "--steps", "1", "--warmup", "1")
    check_two_ranks(d, n)
    assert d["value"] > 0 and d["steps"] == 1 and "cpu_baseline" not in d
    assert d["config"]["sharded_impl"].startswith("bt_mgpu")
    import numpy as np
    from boxtree_amd import HIPArrayContext, TreeBuilder
    actx = HIPArrayContext(0)
    # (the recipe draws x, then y, then z: make_workload_numpy)
    chunks = []
    for r in range(2):
        rng = np.random.default_rng(15 + r)
        chunks.append([rng.random(n) for _ in range(3)])
    pts = [np.concatenate([chunks[0][ax], chunks[1][ax]]) for ax in range(3)]
    tree, _ = TreeBuilder(actx)(actx, [actx.from_numpy(p) for p in pts], max_particles_in_box=64)
    assert d["config"]["global_nboxes"] == int(tree.nboxes)
    assert d["config"]["nlevels"] == int(tree.nlevels)


def test_step_roofline_arithmetic():
    """roofline.step of the bench line: c3's figures give the 31 GB / 0.23 of peak of DESIGN.md."""
    sys.path.insert(0, ROOT)
    import bench
    r = bench.step_roofline(10**8, 5261406, 4, 16.0, 356782353, 16.45)
    assert r["algorithmic_bytes_per_particle"] == 24 + 64 + 8 + 64 + 12 + 16 + 60 + 24
    assert abs(r["algorithmic_bytes"] - (272e8 + 214 * 5261406 + 4 * 356782353)) < 1
    assert 0.22 < r["frac"] < 0.25 and r["unit"] == "GB/s"
    assert bench.step_roofline(0, 0, 0, 16.0, 0, 0.0)["frac"] == 0.0
