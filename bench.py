#!/usr/bin/env python
"""Benchmark of the hot path: TreeBuilder + FMMTraversalBuilder on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload c3] [--n N_POINTS]

A "step" is one full pass of the hot path (tree build + traversal build) over
one batch of synthetic particles already resident in HBM.  Rank 0 prints ONE
JSON line (see DESIGN.md section "Measurement").

Workloads (BASELINE.json configs):
  c2  3D uniform, 10^7 sources=targets, max_particles_in_box=64
  c3  3D sphere surface, 10^8 points, max_particles_in_box=64   (default: the
      configuration the metric "3D 10^8 pts" is quoted on)
  c3c the clustered variant of c3: z -> sign(z)|z|^(1/4), renormalised
  c1  2D uniform, 10^5 points, max_particles_in_box=30 (the reference's CPU-runnable case)
  c4  3D 10^8 sources + 10^7 targets with target radii, stick_out_factor=0.25
  c5  3D uniform, 1.25*10^8 points per rank drawn with default_rng(15 + rank): BASELINE
      configs[4] (10^9 points over 8 GPUs); the default for --gpus N > 1
For --gpus N > 1 every rank holds its own chunk of the workload (weak scaling):
global bounding box by RCCL all-reduce, particles exchanged all-to-all by
top-level Morton cell, then every rank builds the subtrees it owns, numbers them
globally and builds the lists of its own boxes on a local essential tree.

Launching.  Under a launcher (torch.distributed.run sets RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_*) the script is one rank.  Without one, ``--gpus N`` with N > 1 starts the N ranks
itself (one child process per GPU, rendezvous on 127.0.0.1) and rank 0 prints the line.
Ranks that have to share a GPU (fewer devices than ranks: a test setup, not a
measurement) talk over gloo instead of RCCL, which refuses two ranks on one device.
``--dry-run`` does the rendezvous and the particle exchange only (CPU tensors over gloo
when there is no GPU): it checks the launcher, it measures nothing.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak, MI355X_MICROARCH.md


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=None, choices=["c1", "c2", "c3", "c3c", "c4", "c5"],
                    help="default: c3 on one GPU, c5 (BASELINE configs[4]) for --gpus N > 1")
    ap.add_argument("--backend", default=None, choices=["nccl", "gloo"],
                    help="collectives (default: nccl = RCCL; gloo if ranks share a GPU or there is none)")
    ap.add_argument("--dry-run", action="store_true",
                    help="rendezvous + particle exchange only; no tree, no measurement")
    ap.add_argument("--rng", default="numpy", choices=["numpy", "torch"],
                    help="numpy: SURVEY 8d's host recipes (np.random.default_rng), uploaded; "
                         "torch: the same distributions drawn on the device")
    # (--points-per-gpu: `--n` is read as an abbreviation of its own options by
    # torch.distributed.run's parser, even after the script name)
    ap.add_argument("--n", "--points-per-gpu", dest="n", type=int, default=None,
                    help="override particle count per GPU")
    ap.add_argument("--cpu-sample", type=int, default=16_000_000,
                    help="particles in the CPU-baseline sample (0 disables)")
    ap.add_argument("--mpb", type=int, default=None)
    ap.add_argument("--force-dist", action="store_true",
                    help="run the N>1 code path (exchange, numbering, gather) even with one rank")
    return ap.parse_args()


# {{{ synthetic data (on device)

def make_workload(torch, device, workload, n, seed):
    """Returns dict(particles=[x,y,z], targets=None|[...], target_radii=None|t, kw=...)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    f64 = torch.float64
    if workload in ("c2", "c5"):
        # c5 = BASELINE configs[4]: 10^9 uniform points over 8 GPUs, 1.25*10^8 per rank
        # (rank g draws its chunk with seed 15+g)
        n = n or (10**7 if workload == "c2" else 125 * 10**6)
        pts = [torch.rand(n, generator=g, dtype=f64, device=device) for _ in range(3)]
        return dict(name="3D uniform random, sources=targets", n=n, particles=pts,
                    targets=None, kw={})
    if workload == "c1":
        n = n or 10**5
        pts = [torch.rand(n, generator=g, dtype=f64, device=device) for _ in range(2)]
        return dict(name="2D uniform random, sources=targets", n=n, particles=pts,
                    targets=None, kw={})
    if workload in ("c3", "c3c"):
        n = n or 10**8
        v = [torch.randn(n, generator=g, dtype=f64, device=device) for _ in range(3)]
        nrm = torch.sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2])
        pts = [(c / nrm).contiguous() for c in v]
        del v, nrm
        name = "3D sphere-surface points, sources=targets"
        if workload == "c3c":
            # polar-cap concentration: z -> sign(z)|z|^(1/4), renormalised
            z = torch.sign(pts[2]) * torch.abs(pts[2]) ** 0.25
            nrm = torch.sqrt(pts[0] * pts[0] + pts[1] * pts[1] + z * z)
            pts = [(pts[0] / nrm).contiguous(), (pts[1] / nrm).contiguous(),
                   (z / nrm).contiguous()]
            del z, nrm
            name = "3D sphere surface clustered towards the poles, sources=targets"
        return dict(name=name, n=n, particles=pts, targets=None, kw={})
    if workload == "c4":
        n = n or 10**8
        nt = max(n // 10, 1)
        src = [torch.rand(n, generator=g, dtype=f64, device=device) for _ in range(3)]
        tgt = [torch.rand(nt, generator=g, dtype=f64, device=device) for _ in range(3)]
        radii = (2.0 ** (-10.0 * torch.rand(nt, generator=g, dtype=f64, device=device))
                 * 2.0 ** -7)
        return dict(name="3D uniform sources + 10% targets with radii", n=n + nt,
                    particles=src, targets=tgt,
                    kw=dict(target_radii=radii, stick_out_factor=0.25))
    raise ValueError(workload)


def make_workload_numpy(workload, n, seed):
    """Same recipes on the host for the CPU baseline sample."""
    rng = np.random.default_rng(seed)
    if workload in ("c2", "c5"):
        return dict(particles=[rng.random(n) for _ in range(3)], targets=None, kw={})
    if workload == "c1":
        return dict(particles=[rng.random(n) for _ in range(2)], targets=None, kw={})
    if workload in ("c3", "c3c"):
        v = rng.standard_normal((3, n))
        v /= np.sqrt((v * v).sum(axis=0))
        if workload == "c3c":
            v[2] = np.sign(v[2]) * np.abs(v[2]) ** 0.25
            v /= np.sqrt((v * v).sum(axis=0))
        return dict(particles=[np.ascontiguousarray(v[i]) for i in range(3)],
                    targets=None, kw={})
    if workload == "c4":
        # sources default_rng(15), targets default_rng(16), radii
        # 2**default_rng(12).uniform(-10, 0) * 2^-7
        nt = max(n // 10, 1)
        trng = np.random.default_rng(seed + 1)
        return dict(particles=[rng.random(n) for _ in range(3)],
                    targets=[trng.random(nt) for _ in range(3)],
                    kw=dict(target_radii=2.0 ** np.random.default_rng(12).uniform(-10, 0, nt)
                            * 2.0 ** -7,
                            stick_out_factor=0.25))
    raise ValueError(workload)


# N > 1 steps produce the global user ids of the received particles (BOXTREE_HIP_BENCH_IDS=0: a
# step without them, for comparison with round 4's lines)
WITH_IDS = os.environ.get("BOXTREE_HIP_BENCH_IDS", "1") != "0"
WORKLOAD_MPB = {"c1": 30}      # max_particles_in_box of a workload (64 unless listed)

# }}}


def usable_cores():
    """Host cores this process may actually use: the affinity mask, capped by the
    container's CPU quota (cgroup cpu.max) -- a GPU box of the pool shows 256 cores and
    grants 16; 256 threads on 16 cores run the OpenMP oracle ten times slower than 16."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, -(-q // per)))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(workload, n_sample, mpb, reps=3):
    """The CPU oracle ("port" of the reference's level-loop algorithm) timed on bounded
    samples of the same workload: once with one thread, once with all host cores (the
    same source built with -fopenmp, identical results), best of `reps` runs each.
    Reported, never the target."""
    from oracle import oracle

    def run(variant, threads, n):
        nthreads = oracle.set_variant(variant, threads)
        w = make_workload_numpy(workload, n, 15)
        nn = len(w["particles"][0]) + (len(w["targets"][0]) if w["targets"] else 0)
        times = []
        for _ in range(reps):
            t0 = time.perf_counter()
            tree = oracle.build_tree(w["particles"], targets=w["targets"],
                                     max_particles_in_box=mpb, **w["kw"])
            oracle.build_traversal(tree)
            times.append(time.perf_counter() - t0)
        return nn, nthreads, times

    ncores = usable_cores()
    try:
        # one thread: a quarter of the sample keeps three repetitions within ~20 s
        n1, _, t1 = run("seq", None, max(n_sample // 4, 1))
        nall, nthreads, tall = run("omp", ncores, n_sample)
    finally:
        oracle.set_variant("seq")
    fmt = lambda ts: "/".join(f"{t:.2f}" for t in ts)  # noqa: E731
    return {
        "value": nall / min(tall), "unit": "particles/s", "cores": nthreads, "kind": "port",
        "sample": f"{workload} recipe at {nall} particles, tree build + traversal, "
                  f"oracle/boxtree_oracle.c built with -fopenmp on {nthreads} threads "
                  f"(all cores this process may use: {os.cpu_count()} present, CPU quota "
                  f"{ncores}), best of {reps} runs ({fmt(tall)} s)",
        "single_thread": {
            "value": n1 / min(t1), "unit": "particles/s", "cores": 1,
            "sample": f"{workload} recipe at {n1} particles, sequential build of the same "
                      f"source, best of {reps} runs ({fmt(t1)} s)",
        },
    }


WORKLOAD_NAMES = {
    "c1": "2D uniform random, sources=targets",
    "c2": "3D uniform random, sources=targets",
    "c5": "3D uniform random, sources=targets",
    "c3": "3D sphere-surface points, sources=targets",
    "c3c": "3D sphere surface clustered towards the poles, sources=targets",
    "c4": "3D uniform sources + 10% targets with radii",
}
WORKLOAD_N = {"c1": 10**5, "c2": 10**7, "c3": 10**8, "c3c": 10**8, "c4": 10**8, "c5": 125 * 10**6}


def make_workload_uploaded(torch, device, workload, n, seed):
    """SURVEY 8d's recipes as written (np.random.default_rng(seed) on the host), uploaded
    one array at a time: the inputs the parity tests use at oracle size, at full size."""
    n = n or WORKLOAD_N[workload]
    w = make_workload_numpy(workload, n, seed)

    def up(a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(device)

    particles = [up(a) for a in w["particles"]]
    w["particles"] = None
    targets = None
    if w["targets"] is not None:
        targets = [up(a) for a in w["targets"]]
    kw = {k: (up(v) if isinstance(v, np.ndarray) else v) for k, v in w["kw"].items()}
    nn = len(particles[0]) + (len(targets[0]) if targets is not None else 0)
    return dict(name=WORKLOAD_NAMES[workload], n=nn, particles=particles, targets=targets, kw=kw)


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n_ranks):
    """`--gpus N` without a launcher's environment: start the N ranks (this script, one
    child per GPU, LOCAL_RANK = RANK, rendezvous on 127.0.0.1), let rank 0's stdout through
    (the single JSON line) and the other ranks' to stderr.  The first rank that fails takes
    the others down (by their own PIDs)."""
    import subprocess
    port = free_port()
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks),
                   LOCAL_WORLD_SIZE=str(n_ranks), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), BOXTREE_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]],
                                      env=env, stdout=None if r == 0 else sys.stderr))
    rc = 0
    live = set(range(n_ranks))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code
                print(f"bench.py: rank {r} exited with status {code}; stopping the others",
                      file=sys.stderr)
                for q in live:
                    procs[q].terminate()
        time.sleep(0.05)
    return rc


def main():
    args = parse_args()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))
    import torch
    import torch.distributed as dist

    world = int(env_world or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s); "
              f"reporting n_gpus = {world}", file=sys.stderr)
    if args.workload is None:
        args.workload = "c5" if world > 1 else "c3"
    if args.mpb is None:
        args.mpb = WORKLOAD_MPB.get(args.workload, 64)
    distributed = world > 1 or args.force_dist
    have_gpu = torch.cuda.is_available()
    if not have_gpu and not args.dry_run:
        raise RuntimeError("bench.py needs a HIP device: the hot path has no CPU fallback "
                           "(--dry-run checks the launcher and the exchange plumbing only)")
    ndev = torch.cuda.device_count() if have_gpu else 0
    shared_gpu = have_gpu and ndev < local_world
    dev_index = local_rank % ndev if have_gpu else None
    backend = args.backend or ("nccl" if have_gpu and not shared_gpu else "gloo")
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if have_gpu:
            torch.cuda.set_device(dev_index)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group("gloo")
    device = torch.device("cuda", dev_index) if have_gpu else torch.device("cpu")
    if have_gpu:
        torch.cuda.set_device(device)

    if args.dry_run:
        return dry_run(args, torch, dist, device, world, rank, backend, distributed)

    from boxtree_amd import FMMTraversalBuilder, HIPArrayContext, TreeBuilder
    from boxtree_amd import _lib

    actx = HIPArrayContext(dev_index)
    tb = TreeBuilder(actx)
    tg = FMMTraversalBuilder(actx)

    if args.rng == "numpy":
        w = make_workload_uploaded(torch, device, args.workload, args.n, 15 + rank)
    else:
        w = make_workload(torch, device, args.workload, args.n, 15 + rank)
    n_local = w["n"]
    build_kw = dict(w["kw"])
    particles, targets = w["particles"], w["targets"]

    torch.cuda.synchronize()
    xinfo = {}

    # N > 1 over RCCL, point particles (sources, and separate point targets): the library's
    # own multi-GPU entries, targets with extents included.  Other process groups (gloo: ranks
    # that share a GPU) and refine weights go through the torch implementation of the same steps.
    native_comm = None
    # (build keywords the entries know: targets with extents)
    native_kw_ok = set(build_kw) <= {"target_radii", "stick_out_factor", "extent_norm"}
    if (distributed and backend == "nccl" and native_kw_ok
            and os.environ.get("BOXTREE_HIP_NATIVE_MGPU", "1") != "0"):
        # (no fallback: a job on RCCL that cannot make its communicator ends here, so that a
        # timing can never silently be of the other implementation)
        from boxtree_amd.distributed import native as nat
        native_comm = nat.rccl_comm(actx, dist)
    elif (distributed and shared_gpu and backend == "gloo" and native_kw_ok and world > 1
            and os.environ.get("BOXTREE_HIP_NATIVE_MGPU", "1") != "0"):
        # ranks that share a GPU (a test setup): the same bt_mgpu_* entries over the library's
        # shared-memory communicator -- every byte crosses the host, nothing here is a scaling
        # measurement, but it is the N > 1 code path and not its torch twin
        from boxtree_amd.distributed import native as nat
        box = [f"/bt_bench_{os.getpid()}" if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        native_comm = nat.shm_comm(box[0], rank, world)
    # which implementation of the sharded build this job times (config.sharded_impl)
    sharded_impl = None
    if distributed:
        sharded_impl = "torch"
        if native_comm is not None:
            sharded_impl = "bt_mgpu" + (" over its shared-memory communicator (ranks share a GPU)"
                                        if native_comm.kind == "processes" else "")

    stage_acc: dict[str, float] = {}
    sort_ms = []
    info = {}
    last_exchange = {}

    golden_c5 = None
    if args.workload == "c5" and n_local == WORKLOAD_N["c5"] and args.mpb == 64 and args.rng == "numpy":
        try:
            golden_c5 = json.load(open(os.path.join(ROOT, "tests", "golden", "c5_global_counts.json")))
            golden_c5 = golden_c5["worlds"].get(str(world))
        except (OSError, KeyError, ValueError):
            golden_c5 = None
    c5_local = {}

    def c5_record(global_box_ids, counts_cumul, nboxes, nlevels, level_starts, ids_in_tree_order=None,
                  first_position=0):
        """This rank's share of the tree checksum (linear in the counts: the ranks' shares add
        up to the single-GPU tree's, boxtree_amd/distributed/checksum.py) and of the checksum of
        the particle order (global user id at every tree position of the rank's slice)."""
        from boxtree_amd.distributed.checksum import particle_order_checksum, tree_checksum
        c5_local.update(checksum=tree_checksum(torch, global_box_ids, counts_cumul), nboxes=int(nboxes),
                        nlevels=int(nlevels), level_start_box_nrs=[int(v) for v in level_starts])
        if ids_in_tree_order is not None:
            c5_local["ids_checksum"] = particle_order_checksum(torch, ids_in_tree_order, first_position)

    def step(instrumented=False):
        """One pass of the path.  The timed steps read only the sort's event times (the
        sort of a step has long completed when its build returns); the per-stage times
        would make the host wait for the end of the stream-ordered tails, so they come
        from extra, untimed steps after the timed region (instrumented=True)."""
        p_, t_, kw_ = particles, targets, build_kw
        xs = None
        if native_comm is not None:
            # steps 1-6 behind the C ABI (bt_mgpu_*): exchange, build, numbering, local
            # essential tree, lists of the rank's own boxes
            from boxtree_amd.distributed import native as nat
            if targets is None:
                p_, kw_, xs = nat.exchange_particles(actx, native_comm, particles, args.mpb)
                t_ = None
            elif "target_radii" in build_kw:
                # targets with extents: a target that sticks out of the shared top boxes stays in
                # one of them and travels to the owner of its first cell; radii ride along
                p_, t_, r_, kw_, xs = nat.exchange_particles(
                    actx, native_comm, particles, args.mpb, targets=targets,
                    target_radii=build_kw["target_radii"],
                    stick_out_factor=build_kw.get("stick_out_factor"),
                    extent_norm=build_kw.get("extent_norm", "linf"))
                kw_ = dict(kw_, target_radii=r_)
            else:
                # separate point targets travel to the owners of their cells like the sources
                p_, t_, kw_, xs = nat.exchange_particles(actx, native_comm, particles, args.mpb,
                                                         targets=targets)
            tree, _ = tb(actx, p_, targets=t_, max_particles_in_box=args.mpb, **kw_)
            st = _lib.SortStats()
            actx.lib.bt_get_sort_stats(actx.handle, st)
            # particle identity (part of the step: a Tree comes with user_source_ids that name the
            # caller's particles): global user ids of what this rank received, over the exchange's
            # kept plan -- 4 more bytes per particle that changed rank, in messages of their own
            route = xs["route"]
            id_bytes = 0
            gids = None
            if WITH_IDS:
                gids = route.global_ids("sources")
                # (int32 = the reference's particle_id_t while the global count fits, else int64)
                id_bytes = gids.element_size() * route.n_sent["sources"]
                if t_ is not None:
                    tgids = route.global_ids("targets")
                    id_bytes += tgids.element_size() * route.n_sent["targets"]
                    del tgids
            last_exchange.update(bytes_sent=int(xs["bytes_sent"]), owned=int(len(p_[0])),
                                 a2a_ms=xs["a2a_ms"], id_bytes=id_bytes)
            num = nat.number_sharded_tree(actx, native_comm, tree)
            gtree, let = nat.build_local_essential_tree(actx, native_comm, tree, num)
            trav, _ = tg(actx, gtree, _target_boxes_mask=let["target_boxes_mask"],
                         _active_level_ranges=let["active_level_ranges"])
            xinfo.update(let_nboxes_rank0=int(let["nboxes"]),
                         halo_boxes_received_rank0=int(let["halo_boxes_received"]),
                         global_nboxes=int(num["nboxes"]),
                         sharded_traversal="bt_mgpu_* entries: local essential tree (halo of "
                                           "neighbouring cells); lists for own boxes + shared top levels")
            nboxes, nlevels = int(num["nboxes"]), int(gtree.nlevels)
            if instrumented and golden_c5 is not None:
                c5_record(num["box_ids"], tree.box_source_counts_cumul, nboxes, num["nlevels"],
                          num["global_level_start_box_nrs"],
                          None if gids is None else gids[tree.user_source_ids.long()], num["source_offset"])
            return finish_step(st, trav, nboxes, nlevels, instrumented)
        if distributed:
            # the exchange is part of the path (and of the timed step) for N > 1
            from boxtree_amd.distributed import exchange_particles
            p_, t_, kw_, xs = exchange_particles(
                actx, dist, particles, targets, build_kw, max_particles_in_box=args.mpb)
            id_bytes = 0
            if WITH_IDS:
                # (as on the bt_mgpu_* path: global user ids of the received particles)
                for which, arrs in (("sources", particles), ("targets", targets)):
                    if arrs is not None:
                        g_ = xs["route"].global_ids(which)
                        id_bytes += 4 * (len(arrs[0]) - xs["route"]._get(which)["s_split"][rank])
                        del g_
            last_exchange.update(bytes_sent=int(xs["bytes_sent"]), owned=int(len(p_[0])),
                                 events=xs.get("a2a_events", []), id_bytes=id_bytes)
        tree, _ = tb(actx, p_, targets=t_, max_particles_in_box=args.mpb, **kw_)
        st = _lib.SortStats()
        actx.lib.bt_get_sort_stats(actx.handle, st)
        if distributed and xs["plan"] is not None:
            # global box numbers, box arrays of all ranks, then the interaction lists
            # of this rank's boxes (cross-boundary lists included)
            from boxtree_amd.distributed import (build_local_essential_tree,
                                                 gather_global_box_tree, number_sharded_tree)
            num = number_sharded_tree(dist, tree, xs)
            if os.environ.get("BOXTREE_HIP_SHARDED", "let") == "gather":
                # every rank gets ALL box arrays (simple, not scalable)
                gtree = gather_global_box_tree(actx, dist, tree, num)
                mask, ranges = num["target_boxes_mask"], num["active_level_ranges"]
                how = "global box arrays all-gathered"
            else:
                # local essential tree: shared top levels + own subtrees + halo subtrees
                gtree, let = build_local_essential_tree(actx, dist, tree, xs, num)
                mask, ranges = let["target_boxes_mask"], let["active_level_ranges"]
                how = "local essential tree (halo of neighbouring cells)"
                xinfo.update(let_nboxes_rank0=int(let["nboxes"]),
                             halo_boxes_received_rank0=int(let["halo_boxes_received"]))
            trav, _ = tg(actx, gtree, _target_boxes_mask=mask, _active_level_ranges=ranges)
            xinfo.update(global_nboxes=int(num["nboxes"]), sharded_traversal=how
                         + "; lists for own boxes + shared top levels")
            nboxes, nlevels = int(num["nboxes"]), int(gtree.nlevels)
        else:
            trav, _ = tg(actx, tree)
            nboxes, nlevels = int(tree.nboxes), int(tree.nlevels)
            if instrumented and golden_c5 is not None and not distributed:
                c5_record(torch.arange(nboxes, device=device), tree.box_source_counts_cumul, nboxes,
                          nlevels, actx.to_numpy(tree.level_start_box_nrs), tree.user_source_ids)
        return finish_step(st, trav, nboxes, nlevels, instrumented)

    def finish_step(st, trav, nboxes, nlevels, instrumented):
        times = {}
        if instrumented:
            times = dict(tb.last_stage_times)       # build and traversal stages
        info.update(nboxes=nboxes, nlevels=nlevels,
                    n_list1=int(trav.neighbor_source_boxes_lists.shape[0]),
                    n_list2=int(trav.from_sep_siblings_lists.shape[0]),
                    n_colleagues=int(trav.same_level_non_well_sep_boxes_lists.shape[0]),
                    n_list3=int(sum(int(b.count) for b in trav.from_sep_smaller_by_level)),
                    n_list4=int(trav.from_sep_bigger_lists.shape[0]),
                    n_close=(0 if trav.from_sep_close_smaller_lists is None else
                             int(trav.from_sep_close_smaller_lists.shape[0])
                             + int(trav.from_sep_close_bigger_lists.shape[0])))
        return st, times

    actx.set_stage_timing(False)      # ~30 event records per step; see step()
    if native_comm is not None and world > 1:
        # One untimed step through the library's own RCCL entries before anything is measured,
        # and an agreement over torch's process group that it worked on every rank: if it did
        # not, every rank ends the job (bench.py does not move a job to the torch implementation:
        # BOXTREE_HIP_NATIVE_MGPU=0 selects that one, and the line then says so).
        ok, why = 1, ""
        try:
            step()
            actx.synchronize()
        except (RuntimeError, OSError, ValueError) as e:
            ok, why = 0, str(e)
            print(f"bench.py: rank {rank}: the bt_mgpu_* path failed ({e})", file=sys.stderr)
        flag = torch.tensor([ok], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            raise RuntimeError("bench.py: the sharded build through bt_mgpu_* failed on some rank"
                               + (f" (this one: {why})" if why else ""))
    for _ in range(args.warmup):
        step()

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        st, times = step()
        sort_ms.append((st.full_pass_ms_avg, st.passes, st.n, st.first_pass_ms,
                        st.first_pass_identity, st.full_passes, st.bytes_per_element_per_pass,
                        st.digit_bits))
    barrier()
    elapsed = time.perf_counter() - t0
    actx.synchronize()                # a device-side failure of the last step raises here
    n_instrumented = 3
    actx.set_stage_timing(True)
    for _ in range(n_instrumented):
        _, times = step(instrumented=True)
        for k, v in times.items():
            stage_acc[k] = stage_acc.get(k, 0.0) + v
    barrier()
    if golden_c5 is not None and c5_local:
        # the global numbering this run arrived at against the tree ONE GPU builds from the
        # chunks of all ranks (tests/golden/c5_global_counts.json, tools/c5_full.py)
        from boxtree_amd.distributed.checksum import wrap_int64
        total = c5_local["checksum"]
        ids_total = c5_local.get("ids_checksum")
        if distributed and world > 1:
            shares = [None] * world
            dist.all_gather_object(shares, (c5_local["checksum"], c5_local.get("ids_checksum")))
            total = wrap_int64(sum(s_[0] for s_ in shares))
            ids_total = (wrap_int64(sum(s_[1] for s_ in shares))
                         if all(s_[1] is not None for s_ in shares) else None)
        same = (total == golden_c5["counts_cumul_checksum"] and c5_local["nboxes"] == golden_c5["nboxes"]
                and c5_local["nlevels"] == golden_c5["nlevels"]
                and c5_local["level_start_box_nrs"] == golden_c5["level_start_box_nrs"])
        ids_same = None
        if ids_total is not None and "user_source_ids_checksum" in golden_c5:
            # every particle at its place of the global tree order, named by its global user id
            ids_same = ids_total == golden_c5["user_source_ids_checksum"]
            same = same and ids_same
        xinfo["c5_check"] = {
            "matches_single_gpu_tree": bool(same), "nboxes": c5_local["nboxes"],
            "nlevels": c5_local["nlevels"], "counts_cumul_checksum": total,
            "user_source_ids_checksum": ids_total, "particle_order_matches": ids_same,
            "expected": {k: golden_c5[k] for k in ("nboxes", "nlevels", "counts_cumul_checksum")},
            "source": f"tests/golden/c5_global_counts.json, worlds[{world}]: the tree one GPU builds from "
                      f"the {world} chunk(s) default_rng(15..{14 + world}), checked there with the "
                      "reference's tree assertions",
        }
        if not same and rank == 0:
            print("bench.py: c5_check FAILED: " + json.dumps(xinfo["c5_check"]), file=sys.stderr)
    if distributed:
        xinfo.update(exchange_report(torch, dist, device, world, elapsed, n_local, last_exchange,
                                     backend, shared_gpu))
        elapsed = xinfo.pop("_elapsed_max")
        n_total = xinfo.pop("_n_total")
    else:
        n_total = n_local

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = n_total * args.steps / elapsed
        pass_ms = float(np.mean([s[0] for s in sort_ms]))
        n_sorted = sort_ms[-1][2]
        # bytes a digit pass reads + writes per particle: 24 for (u64 key, u32 id) pairs, 16
        # when the build packs the id under the path bits of one 64-bit word (point
        # particles; DESIGN.md section 3, LAB_NOTES.md "Packed keys")
        pass_bytes = float(sort_ms[-1][6] or 24)
        keys_only = pass_bytes == 16.0
        achieved = pass_bytes * n_sorted / (pass_ms * 1e-3) / 1e9 if pass_ms > 0 else 0.0
        # HBM bytes per launch from the PMC counters: they cannot be collected in this
        # process (separate rocprofv3 --pmc passes, corrected as MI355X_MICROARCH.md
        # prescribes: tools/pmc_onesweep.sh), so the figure is read from the committed
        # summary of those passes -- when it was taken at this N -- and its file is named
        traffic, traffic_source = None, None
        try:
            pmc_files = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles"))
                               if f.endswith("_pmc_onesweep_keys.json" if keys_only
                                             else "_pmc_onesweep.json"))
            pmc = json.load(open(os.path.join(ROOT, "profiles", pmc_files[-1])))
            if int(pmc["n_pairs"]) == int(n_sorted):
                traffic = pmc["traffic_bytes_per_launch"]
                traffic_source = (f"profiles/{pmc_files[-1]} (separate rocprofv3 --pmc FETCH_SIZE / "
                                  f"WRITE_SIZE passes over tools/sort_bench.py at n = {n_sorted}; "
                                  "not measured in this run)")
            else:
                traffic_source = (f"null: profiles/{pmc_files[-1]} was taken at n = {pmc['n_pairs']}, "
                                  f"this run sorts {n_sorted} pairs")
        except (OSError, KeyError, ValueError, IndexError) as e:
            traffic_source = f"null: no PMC summary under profiles/ ({type(e).__name__})"
        # what a plain device-to-device copy of the same byte count reaches on this GPU,
        # measured here: the practical ceiling the sort pass is to be read against
        copy_gbps = None
        try:
            nbytes = int(pass_bytes / 2 * n_sorted)
            src_b = torch.empty(nbytes, dtype=torch.uint8, device=device)
            dst_b = torch.empty_like(src_b)
            dst_b.copy_(src_b)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                dst_b.copy_(src_b)
            e1.record()
            torch.cuda.synchronize()
            copy_gbps = 2.0 * nbytes * 5 / (e0.elapsed_time(e1) * 1e-3) / 1e9
            del src_b, dst_b
        except RuntimeError:
            pass
        rng_note = ("np.random.default_rng(15 + rank) on the host, uploaded (SURVEY 8d recipe)"
                    if args.rng == "numpy" else "torch.Generator(15 + rank) on the device")
        out = {
            "metric": "particles/sec tree build+traversal, 3D 10^8 pts; radix-sort HBM GB/s vs peak",
            "value": value,
            "unit": "particles/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64 coordinates, u64 Morton keys, int32 ids",
            "data": "synthetic",
            "config": {
                "workload": f"{args.workload}: {w['name']}, {n_local} particles per GPU, "
                            f"max_particles_in_box={args.mpb}, kind=adaptive; inputs: {rng_note}",
                "nboxes": info.get("nboxes"), "nlevels": info.get("nlevels"),
                "list1_entries": info.get("n_list1"), "list2_entries": info.get("n_list2"),
                "parallelism": f"{world} rank(s), one per GPU, shard by top-level Morton cell",
                **({"sharded_impl": sharded_impl} if sharded_impl else {}),
                **({"scaling_note": "weak scaling of BASELINE configs[4] (1.25e8 uniform points per rank): "
                                    "the matching one-GPU figure is `bench.py --gpus 1 --workload c5`; the "
                                    "default N = 1 line measures configs[2] (10^8 sphere-surface points)"}
                   if world > 1 and args.workload == "c5" else {}),
                **({"one_gpu_reference": one_gpu_reference(args.workload)} if world > 1 else {}),
                **xinfo,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": (f"bt::onesweep_keys_kernel<..., {sort_ms[-1][7]}> (one {sort_ms[-1][7]}-bit "
                           "digit pass of the tree build's sort: ONE 64-bit word per particle, "
                           "Morton path bits over the user id, read and written once: 16 bytes "
                           "per particle)" if keys_only else
                           "bt::onesweep_kernel<unsigned long, ..., false> (one 8-bit digit pass of "
                           "the 64-bit Morton-key sort that reads and writes keys and values: "
                           "24 bytes per pair; the first pass synthesises its values, moves "
                           "20 bytes per pair and is reported apart)"),
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": traffic_source,
                "algorithmic_bytes_per_launch": pass_bytes * n_sorted,
                "algorithmic_bytes_per_particle": pass_bytes,
                "digit_bits": sort_ms[-1][7],
                "device_copy_GBps_same_bytes": copy_gbps,
                "avg_launch_ms": pass_ms,
                "passes_per_sort": sort_ms[-1][1],
                "launches_averaged_per_sort": sort_ms[-1][5],
                "first_pass": {
                    "synthesised_values": bool(sort_ms[-1][4]),
                    "avg_launch_ms": float(np.mean([s[3] for s in sort_ms])),
                    "algorithmic_bytes_per_launch": (20.0 if sort_ms[-1][4] else pass_bytes) * n_sorted,
                },
            },
            "stages_ms": {k: v / n_instrumented for k, v in stage_acc.items()},
        }
        nb_ = int(info.get("nboxes") or 0) if world == 1 else int(xinfo.get("let_nboxes_rank0") or 0)
        entries = sum(int(info.get(k, 0) or 0) for k in (
            "n_colleagues", "n_list1", "n_list2", "n_list3", "n_list4", "n_close"))
        out["roofline"]["step"] = step_roofline(n_sorted, nb_, int(sort_ms[-1][1]), pass_bytes, entries,
                                                ms_per_step)
        # list output rate of the traversal stages (the walks are latency /
        # L2 bound; what they deliver is 4 bytes per list entry)
        st_ms = out["stages_ms"]

        def rate(entries, *stages):
            ms = sum(st_ms.get(k, 0.0) for k in stages)
            return {"entries": int(entries), "ms": ms,
                    "output_GBps": (4.0 * entries / (ms * 1e-3) / 1e9) if ms > 0 else None}

        trav_stages = [k for k in st_ms if k.startswith("trav:")]
        out["traversal_output"] = {
            "colleagues+list2": rate(info.get("n_colleagues", 0) + info.get("n_list2", 0),
                                     "trav:colleague rows", "trav:colleagues+list2",
                                     "trav:colleagues", "trav:list2"),
            "lists 1, 3, 4 (+ close)": rate(
                info.get("n_list1", 0) + info.get("n_list3", 0) + info.get("n_list4", 0)
                + info.get("n_close", 0),
                "trav:walk (rows)", "trav:lists 1+3 (final)", "trav:list1 order + list4",
                "trav:list1+list3", "trav:list1", "trav:list3", "trav:list4"),
            "all lists": rate(sum(info.get(k, 0) for k in (
                "n_colleagues", "n_list1", "n_list2", "n_list3", "n_list4", "n_close")),
                *trav_stages),
        }
        # kernel launches and idle GPU time of a step: counted by rocprofv3 (tools/timeline_gaps.py
        # over the kernel trace of this command), not in this process -- the committed
        # timeline of the same workload, named
        try:
            # (the N > 1 code path has a timeline of its own)
            suffix = f"_{args.workload}_forcedist_timeline.txt" if distributed else f"_{args.workload}_timeline.txt"
            tls = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith(suffix))
            first = open(os.path.join(ROOT, "profiles", tls[-1])).readline().split()
            # "launches 104  span 17.530 ms  busy 17.131 ms  idle 0.399 ms"
            out["launches_per_step"] = {"value": int(first[1]), "gpu_idle_ms": float(first[9]),
                                        "source": f"profiles/{tls[-1]} (rocprofv3 --kernel-trace of "
                                                  "bench.py on this workload; not measured in this run)"}
        except (OSError, IndexError, ValueError):
            out["launches_per_step"] = None
        # the CPU baseline is a one-GPU figure (rank 0 at N = 1): beside N ranks it would
        # only lengthen the run
        if args.cpu_sample > 0 and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.workload, args.cpu_sample, args.mpb)
        emit(out)
    if native_comm is not None:
        native_comm.close()
    if distributed:
        dist.destroy_process_group()


def one_gpu_reference(workload):
    """The committed ONE-GPU line of the same workload (same particles per GPU): what a weak-scaling
    efficiency of this N > 1 line is computed against -- value_N / (N * value_1) -- without a second
    run.  `plain`: the single-GPU code path; `sharded_path`: the N > 1 code path on one rank
    (--force-dist: exchange, numbering, local essential tree with nobody to talk to)."""
    ref = {}
    try:
        files = sorted(os.listdir(os.path.join(ROOT, "profiles")))
    except OSError:
        return None
    for key, suffix in (("plain", f"_bench_{workload}.json"), ("sharded_path", f"_bench_{workload}_forcedist.json")):
        cand = [f for f in files if f.endswith(suffix)]
        if not cand:
            continue
        try:
            line = json.loads(open(os.path.join(ROOT, "profiles", cand[-1])).read().strip().splitlines()[-1])
            ref[key] = {"ms_per_step": line["ms_per_step"], "particles_per_s": line["value"],
                        "file": f"profiles/{cand[-1]}"}
        except (OSError, ValueError, KeyError, IndexError):
            continue
    return ref or None


def emit(out):
    # RCCL writes its banner through C stdio, which is block-buffered when stdout
    # is a pipe: push it out first so that the JSON line is the last line
    import ctypes
    ctypes.CDLL(None).fflush(None)
    sys.stdout.flush()
    print(json.dumps(out), flush=True)


def step_roofline(n_particles, nboxes, passes, pass_bytes, list_entries, ms_per_step):
    """The whole step against the HBM roofline (what one rank's build + lists must move at least,
    DESIGN.md section 4 / SURVEY 8d), so that the step's fraction of peak is tracked next to the
    digit pass's.  Per particle of the rank: bounding box 24, key kernel 24 + 8 + 32, digit
    histogram 8, digit passes 16 each (24 for (key, id) pairs), within-leaf order 12, id arrays
    16, coordinate gather 60, box extents 24; per box 190; per list entry 4, per list
    start 4 (six lists)."""
    per_particle = 24 + 64 + 8 + float(pass_bytes) * int(passes) + 12 + 16 + 60 + 24
    step_bytes = per_particle * int(n_particles) + 190.0 * int(nboxes) + 4.0 * int(list_entries) \
        + 4.0 * 6 * int(nboxes)
    achieved = step_bytes / (ms_per_step * 1e-3) / 1e9 if ms_per_step and ms_per_step > 0 else 0.0
    return {
        "algorithmic_bytes": step_bytes,
        "algorithmic_bytes_per_particle": per_particle,
        "achieved": achieved,
        "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS,
        "what": "one rank's build + lists: particles of the rank x per-particle bytes of the build's "
                "kernels + 190 B per box + 4 B per list entry and per list start (exchange kernels of "
                "an N > 1 step not counted: its fraction is a lower bound)",
    }


def exchange_report(torch, dist, device, world, elapsed, n_local, last_exchange, backend,
                    shared_gpu):
    """What the N ranks did, gathered on every rank (collective): the maximum of the timed
    region, the total particle count, the bytes that crossed the links, the device time of
    the payload all-to-all and the owned-particle imbalance after the exchange."""
    a2a_ms = float(last_exchange.get("a2a_ms", 0.0))
    if last_exchange.get("events"):
        a2a_ms = float(sum(e0.elapsed_time(e1) for e0, e1 in last_exchange["events"]))
    mine = torch.tensor([elapsed, float(n_local), float(last_exchange.get("bytes_sent", 0)),
                         float(last_exchange.get("owned", n_local)), a2a_ms,
                         float(last_exchange.get("id_bytes", 0))],
                        dtype=torch.float64, device=device)
    rows = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(rows, mine)
    t = torch.stack(rows).cpu().numpy()
    owned = t[:, 3]
    sent = t[:, 2]
    a2a = t[:, 4]
    rep = {
        "_elapsed_max": float(t[:, 0].max()), "_n_total": int(round(t[:, 1].sum())),
        "collectives": {
            "backend": backend + (" (RCCL %s)" % ".".join(map(str, torch.cuda.nccl.version()))
                                  if backend == "nccl" else
                                  " (ranks share a GPU: not a scaling measurement)" if shared_gpu
                                  else ""),
            "ranks_in_group": int(dist.get_world_size()),
        },
        # coordinates (+ radii / weights) that left their GPU, and the global user ids of those
        # particles (4 bytes each, routed over the same plan in messages of their own)
        "exchange_bytes": int(round(sent.sum() + t[:, 5].sum())),
        "exchange_bytes_coordinates": int(round(sent.sum())),
        "exchange_bytes_ids": int(round(t[:, 5].sum())),
        "exchange_bytes_by_rank": [int(round(v)) for v in sent + t[:, 5]],
        "exchange_a2a_ms_by_rank": [float(v) for v in a2a],
        "owned_particles_by_rank": [int(round(v)) for v in owned],
        "owned_particle_imbalance": float(owned.max() / max(owned.mean(), 1.0)),
    }
    if world > 1 and a2a.max() > 0:
        # every peer sits on its own xGMI link: a rank's payload leaves over world-1 links
        # (bytes and device time of the coordinate all-to-all-v)
        rep["exchange_GBps_per_link"] = float(
            (sent / (world - 1) / np.maximum(a2a, 1e-9) / 1e6).min())
        rep["exchange_GBps_per_gpu"] = float((sent / np.maximum(a2a, 1e-9) / 1e6).min())
    return rep


def dry_run(args, torch, dist, device, world, rank, backend, distributed):
    """Launcher and exchange plumbing without the hot path: every rank draws a small chunk,
    the ranks agree on the root box and the cell owners and trade particles; rank 0 prints
    the line with "dry_run": true and no measurement."""
    n = args.n or 20000
    w = make_workload_numpy(args.workload, n, 15 + rank)
    particles = [torch.from_numpy(a).to(device) for a in w["particles"]]
    xinfo = {}
    if distributed:
        from boxtree_amd.distributed import exchange_particles
        actx = None
        if device.type == "cuda":
            from boxtree_amd import HIPArrayContext
            actx = HIPArrayContext(device.index)
        newp, _, _, xs = exchange_particles(actx, dist, particles, None, {},
                                            max_particles_in_box=args.mpb)
        id_bytes = 0
        if WITH_IDS:
            # global user ids of the received particles, over the exchange's own plan
            gids = xs["route"].global_ids("sources")
            assert len(gids) == len(newp[0])
            id_bytes = 4 * (n - xs["route"]._get("sources")["s_split"][rank])
        xinfo = exchange_report(torch, dist, device, world, 0.0, n,
                                dict(bytes_sent=int(xs["bytes_sent"]), owned=int(len(newp[0])),
                                     events=xs.get("a2a_events", []), id_bytes=id_bytes), backend, False)
        xinfo.pop("_elapsed_max")
        n_total = xinfo.pop("_n_total")
    else:
        n_total = n
    if rank == 0:
        emit({
            "metric": "particles/sec tree build+traversal, 3D 10^8 pts; radix-sort HBM GB/s vs peak",
            "value": None, "unit": "particles/s", "n_gpus": world, "steps": 0, "warmup": 0,
            "ms_per_step": None, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64 coordinates, u64 Morton keys, int32 ids",
            "data": "synthetic", "dry_run": True,
            "config": {"workload": f"{args.workload}: {WORKLOAD_NAMES[args.workload]}, {n} "
                                   f"particles per rank; DRY RUN: rendezvous + particle exchange "
                                   f"only, device {device.type}",
                       "particles_total": n_total, **xinfo},
        })
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
