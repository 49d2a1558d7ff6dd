#!/usr/bin/env python
"""Particle identity across the exchange, the two designs on one GPU (round 5):
(a) the global id rides in the record (one more 8-byte value per f64 particle: measured with the
    weighted exchange, whose records carry exactly one more value), against
(b) records without ids + bt_mgpu_route / bt_mgpu_global_ids over the kept send plan.
One rank (a local communicator): every byte stays on the GPU, so the numbers are the kernels'
cost; what the wire adds is arithmetic (bytes per particle that changes rank).
usage: route_bench.py [n]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from boxtree_amd import HIPArrayContext  # noqa: E402
from boxtree_amd.distributed import native as nat  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10**8
actx = HIPArrayContext(0)
g = torch.Generator(device="cuda")
g.manual_seed(5)
pts = [torch.rand(n, generator=g, dtype=torch.float64, device="cuda") for _ in range(3)]
w = torch.ones(n, dtype=torch.int32, device="cuda")
group = nat.LocalGroup(1)
comm = group.comm(0)


def T():
    torch.cuda.synchronize()
    return time.perf_counter()


def best(fn, reps=5):
    out = []
    for _ in range(reps):
        t0 = T()
        r = fn()
        out.append(1e3 * (T() - t0))
        del r
    return min(out)


t_plain = best(lambda: nat.exchange_particles(actx, comm, pts, 64))
t_wide = best(lambda: nat.exchange_particles(actx, comm, pts, 64, refine_weights=w, max_leaf_refine_weight=64))
p2, kw, stats = nat.exchange_particles(actx, comm, pts, 64)
route = stats["route"]
t_ids = best(lambda: route.global_ids("sources"))
vals = torch.rand(n, generator=g, dtype=torch.float64, device="cuda")
t_fwd8 = best(lambda: route.to_owners(vals))
owned = route.to_owners(vals)
t_rev8 = best(lambda: route.to_callers(owned))
ids = route.global_ids("sources")
ok = bool(torch.equal(route.to_callers(owned), vals)) and bool(
    torch.equal(ids.sort().values, torch.arange(n, dtype=torch.int32, device="cuda")))
print(f"n {n}: exchange (24-B records) {t_plain:.2f} ms; with one more value per record (32 B) {t_wide:.2f} ms "
      f"[includes the weighted job's extra histogram]; global ids over the plan {t_ids:.2f} ms; "
      f"route 8-B values to owners {t_fwd8:.2f} ms, back {t_rev8:.2f} ms; correct {ok}", flush=True)
