"""N>1 host logic on CPU: world_size-2 gloo run of the item-sharded driver (hannoy_amd/multigpu.py)
with a stand-in builder whose selection records are a pure function of the member index — every
rank must end up applying exactly the buffer a single process would have produced."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Batch:
    def __init__(self, first, count, level, stride):
        self.first, self.count, self.level, self.n_layers, self.sel_stride_u64 = first, count, level, level + 1, stride


class FakeBuilder:
    """Follows the hny_builder_* call protocol; search() fills member m's record with f(first+m)."""

    def __init__(self, torch, sizes, stride=5):
        self.torch, self.sizes, self.stride = torch, list(sizes), stride
        self.i, self.first, self.cur = 0, 0, None
        self.applied = []
        self.internal = None
        self.search_calls = []
        self.deferred_calls, self.merged, self.nd = [], [], 0

    def next_batch(self):
        if self.i >= len(self.sizes):
            return Batch(self.first, 0, 0, self.stride)
        self.cur = Batch(self.first, self.sizes[self.i], 0, self.stride)
        self.internal = self.torch.zeros(self.cur.count * self.stride, dtype=self.torch.int64)
        return self.cur

    def _view(self, ptr, words):
        if ptr is None:
            return self.internal
        import ctypes
        arr = np.ctypeslib.as_array((ctypes.c_int64 * words).from_address(ptr))
        return self.torch.from_numpy(arr)

    def search(self, lo, hi, sel_ptr=None):
        self.search_calls.append((self.cur.first, lo, hi))
        buf = self._view(sel_ptr, max(hi, 1) * self.stride)
        for m in range(lo, hi):
            g = self.cur.first + m
            for w in range(self.stride):
                buf[m * self.stride + w] = g * 1000 + w

    def sync(self):
        pass

    def apply(self, sel_ptr=None):
        buf = self._view(sel_ptr, self.cur.count * self.stride)
        self.applied.append(buf[:self.cur.count * self.stride].clone())
        self.first += self.cur.count
        self.i += 1

    # three-step apply (hny_builder_apply_begin / _deferred / _merge): a third of the members are
    # "deferred"; record i of a batch is the pure function g(first, i)
    exch_stride_u64 = 3

    def apply_begin(self, sel_ptr=None):
        buf = self._view(sel_ptr, self.cur.count * self.stride)
        self.applied.append(buf[:self.cur.count * self.stride].clone())
        self.nd = self.cur.count // 3
        return self.nd

    def apply_deferred(self, rank=0, world=1, exch_ptr=None):
        self.deferred_calls.append((self.cur.first, rank, world))
        if exch_ptr is None:
            return
        per = -(-self.nd // world)
        buf = self._view(exch_ptr, world * per * self.exch_stride_u64)
        for i in range(rank, self.nd, world):
            for w in range(self.exch_stride_u64):
                buf[(rank * per + i // world) * self.exch_stride_u64 + w] = self.cur.first * 7919 + i * 10 + w

    def apply_merge(self, exch_ptr=None, rank=0, world=1):
        if exch_ptr is not None:
            per = -(-self.nd // world)
            buf = self._view(exch_ptr, world * per * self.exch_stride_u64)
            recs = []
            for i in range(self.nd):
                at = ((i % world) * per + i // world) * self.exch_stride_u64
                recs.append([int(buf[at + w]) for w in range(self.exch_stride_u64)])
            self.merged.append((self.cur.first, recs))
        self.first += self.cur.count
        self.i += 1

    def run(self):
        n = 0
        while True:
            b = self.next_batch()
            if b.count == 0:
                return n
            self.search(0, b.count)
            self.apply()
            n += 1


SIZES = [1, 1, 2, 3, 7, 64, 129, 130, 255, 1000, 31]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from hannoy_amd.multigpu import Driver
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fb = FakeBuilder(torch, SIZES)
    drv = Driver(fb, torch, dist, rank, world, torch.device("cpu"), min_shard_batch=64)
    n = drv.run()
    digest = [t.numpy().tobytes() for t in fb.applied]
    q.put((rank, n, drv.n_collectives, digest, fb.search_calls, fb.merged, fb.deferred_calls))
    dist.destroy_process_group()


def test_driver_world2_gloo():
    import torch
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # single-process reference
    ref = FakeBuilder(torch, SIZES)
    ref.run()
    want = [t.numpy().tobytes() for t in ref.applied]
    first, exp_merged = 0, []
    for sz in SIZES:  # the deferred share is exchanged when it has >= 32 * world members
        if sz >= 64 and sz // 3 >= 64:
            exp_merged.append((first, [[first * 7919 + i * 10 + w for w in range(3)] for i in range(sz // 3)]))
        first += sz
    for rank, n, ncoll, digest, calls, merged, dcalls in res:
        assert n == len(SIZES)
        assert digest == want, f"rank {rank} applied different selections"
        assert ncoll == sum(1 for s_ in SIZES if s_ >= 64) + len(exp_merged)
        assert merged == exp_merged, f"rank {rank} merged different lists"
        assert all((r, w) == ((rank, 2) if any(f == m[0] for m in exp_merged) else (0, 1)) for f, r, w in dcalls)
    # each sharded batch was split into disjoint contiguous halves covering it
    c0 = {(f, lo, hi) for f, lo, hi in res[0][4]}
    c1 = {(f, lo, hi) for f, lo, hi in res[1][4]}
    assert len(exp_merged) >= 2
    first = 0
    for sz in SIZES:
        if sz >= 64:
            per = -(-sz // 2)
            assert (first, 0, per) in c0 and (first, per, sz) in c1
        else:
            assert (first, 0, sz) in c0 and (first, 0, sz) in c1
        first += sz


def test_shard_arithmetic():
    from hannoy_amd.multigpu import Driver
    for count in (1, 2, 63, 64, 65, 1000, 16384):
        for world in (1, 2, 4, 8):
            got = []
            for r in range(world):
                per, lo, hi = Driver.shard(count, world, r)
                assert hi - lo <= per and lo <= hi <= count
                got += list(range(lo, hi))
            assert got == list(range(count))
