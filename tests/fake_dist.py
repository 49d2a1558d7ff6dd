"""An in-process stand-in for ``torch.distributed`` (test infrastructure): the
ranks are threads of one process sharing one GPU, collectives are implemented
with a barrier and direct tensor copies.  Lets the complete N-rank pipeline of
boxtree_amd/distributed/__init__.py run on a single-GPU box; RCCL itself is not involved."""

import threading


class _ReduceOp:
    SUM, MIN, MAX = "sum", "min", "max"


class FakeWorld:
    def __init__(self, world, backend=None):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world
        # "nccl": the stand-in also offers the list form of all_to_all and names that
        # backend, so that all_to_all_chunked takes the path it takes on RCCL
        self.backend = backend

    def rank_view(self, rank):
        return FakeDist(self, rank)


class FakeDist:
    ReduceOp = _ReduceOp

    def __init__(self, world, rank):
        self._w = world
        self._rank = rank

    def get_world_size(self):
        return self._w.world

    def get_backend(self):
        return self._w.backend

    def __getattr__(self, name):
        # the list form exists only on worlds that play RCCL
        if name == "all_to_all" and self._w.backend == "nccl":
            return self._all_to_all
        raise AttributeError(name)

    def _all_to_all(self, outs, ins):
        self._w.slots[self._rank] = list(ins)
        self._sync()
        for src in range(self._w.world):
            piece = self._w.slots[src][self._rank]
            assert outs[src].shape == piece.shape and outs[src].is_contiguous()
            outs[src].copy_(piece)
        self._sync()

    def get_rank(self):
        return self._rank

    def _sync(self):
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        self._w.barrier.wait()

    def barrier(self):
        self._sync()

    def all_reduce(self, t, op=_ReduceOp.SUM):
        import torch
        self._w.slots[self._rank] = t.clone()
        self._sync()
        stack = torch.stack(list(self._w.slots))
        res = {"sum": stack.sum(0), "min": stack.amin(0), "max": stack.amax(0)}[op]
        self._sync()
        t.copy_(res.to(t.dtype))

    def broadcast(self, t, src=0):
        self._w.slots[self._rank] = t
        self._sync()
        if self._rank != src:
            t.copy_(self._w.slots[src])
        self._sync()

    def broadcast_object_list(self, objs, src=0):
        self._w.slots[self._rank] = objs
        self._sync()
        if self._rank != src:
            objs[:] = list(self._w.slots[src])
        self._sync()

    def all_gather(self, out_list, t):
        self._w.slots[self._rank] = t
        self._sync()
        for r in range(self._w.world):
            out_list[r].copy_(self._w.slots[r])
        self._sync()

    def all_to_all_single(self, out, inp, out_splits=None, in_splits=None):
        world = self._w.world
        if in_splits is None:
            n = inp.shape[0] // world
            in_splits = [n] * world
            out_splits = [out.shape[0] // world] * world
        self._w.slots[self._rank] = (inp, list(in_splits))
        self._sync()
        off = 0
        for src in range(world):
            sinp, ssplits = self._w.slots[src]
            begin = sum(ssplits[:self._rank])
            cnt = ssplits[self._rank]
            assert cnt == out_splits[src]
            out[off:off + cnt].copy_(sinp[begin:begin + cnt])
            off += cnt
        self._sync()
