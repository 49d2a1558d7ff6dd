import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from oracle import oracle
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(f, open(f).read().strip())
    except OSError: pass
w = bench.make_workload_numpy("c3", 4_000_000, 15)
for variant, th in (("seq", None), ("omp", 8), ("omp", 16), ("omp", 32), ("omp", 64), ("omp", 128), ("omp", 256)):
    n = oracle.set_variant(variant, th)
    t0 = time.perf_counter(); tree = oracle.build_tree(w["particles"], max_particles_in_box=64); t1 = time.perf_counter()
    oracle.build_traversal(tree); t2 = time.perf_counter()
    print(variant, n, "tree %.2f trav %.2f" % (t1 - t0, t2 - t1), flush=True)
