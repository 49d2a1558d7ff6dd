#!/usr/bin/env python
"""Send buffer of the exchange for `world` ranks on one GPU: bucket permutation + gather
against the one-sweep partition (bt_partition_pack).  usage: partition_bench.py [n] [world]"""
import ctypes as ct
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from boxtree_amd import HIPArrayContext, _lib  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10**8
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
actx = HIPArrayContext(0)
g = torch.Generator(device="cuda")
g.manual_seed(5)
pts = [torch.rand(n, generator=g, dtype=torch.float64, device="cuda") for _ in range(3)]
ncells = 32768
cells = torch.randint(0, ncells, (n,), generator=g, dtype=torch.int32, device="cuda")
owner = (torch.arange(ncells, device="cuda") * world // ncells).to(torch.int32)
counts = torch.bincount(owner[cells.long()].long(), minlength=world).cpu().tolist()
me = world // 2
s_off = [sum(counts[:k]) for k in range(world + 1)]
ptrs = (ct.c_void_p * 3)(*[p.data_ptr() for p in pts])


def T():
    torch.cuda.synchronize()
    return time.perf_counter()


for rep in range(3):
    send = torch.empty(3 * n, dtype=torch.float64, device="cuda")
    recv = torch.empty(3 * counts[me], dtype=torch.float64, device="cuda")
    perm = torch.empty(n, dtype=torch.int32, device="cuda")
    t0 = T()
    _lib.check(actx.lib.bt_bucket_permutation(actx.handle, ct.c_void_p(cells.data_ptr()), n,
                                              ct.c_void_p(owner.data_ptr()), world,
                                              ct.c_void_p(perm.data_ptr())))
    t1 = T()
    _lib.check(actx.lib.bt_gather_pack(actx.handle, 3, 8, ptrs, ct.c_void_p(perm.data_ptr()), n,
                                       ct.c_void_p(send.data_ptr())))
    t2 = T()
    send2 = torch.empty(3 * n, dtype=torch.float64, device="cuda")
    t3 = T()
    _lib.check(actx.lib.bt_partition_pack(actx.handle, 3, 8, ptrs, ct.c_void_p(cells.data_ptr()), n,
                                          ct.c_void_p(owner.data_ptr()), world, me, s_off[me], 0,
                                          ct.c_void_p(send2.data_ptr()), ct.c_void_p(recv.data_ptr())))
    t4 = T()
    # same records everywhere but in the own segment, which went to recv
    lo, hi = 3 * s_off[me], 3 * s_off[me + 1]
    same = bool(torch.equal(send[:lo], send2[:lo])) and bool(torch.equal(send[hi:], send2[hi:])) \
        and bool(torch.equal(send[lo:hi], recv))
    print(f"world {world} n {n}: permutation {1e3 * (t1 - t0):.2f} ms + gather {1e3 * (t2 - t1):.2f} ms; "
          f"one-sweep partition {1e3 * (t4 - t3):.2f} ms; identical {same}", flush=True)
