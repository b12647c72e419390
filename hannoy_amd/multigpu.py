"""Item-sharded multi-GPU build driver: one process per GPU, every rank holds a full
replica of the vectors and of the graph in HBM.  Per batch each rank runs walk_layer + robust_prune
(hny_builder_search) for its contiguous slice of the batch members against the frozen graph, the
fixed-size selection records are exchanged with ONE all-gather (RCCL over xGMI, `nccl` backend),
and every rank applies the same link ops in the same order (hny_builder_apply), which keeps the
replicas bit-identical.  Most link ops only append to a list and are replayed by every rank; the
targets whose list overflows re-run robust_prune on it (hnsw.rs:547-552, a third of the apply
phase's time): those are split across the ranks as well (hny_builder_apply_begin / _deferred /
_merge) and their finished lists exchanged with a second, small all-gather (272 B per target at
M0=32, a few MB per batch).  Small batches (the ramp-up) are computed redundantly by every rank
instead: the exchange would cost more than the search.

This is the torchrun harness (one PROCESS per GPU, what bench.py --gpus N runs); the same protocol
inside one process — one host thread per GPU, RCCL called directly — is hny_multi.cpp behind
hny_build(n_gpus=N).  On the RCCL path the collectives are issued with the builder's own HIP stream
as torch's current stream, so search -> all-gather -> apply are ordered on the device and the only
host round trip per batch is the deferred-target count.
"""
import contextlib
import math


class Driver:
    def __init__(self, builder, torch, dist, rank=0, world=1, device=None, min_shard_batch=None,
                 host_staged=False, force_collective=False, shard_apply=None, min_shard_deferred=None):
        self.b, self.torch, self.dist = builder, torch, dist
        # host_staged: exchange through host memory (gloo); default is device buffers over RCCL
        self.host_staged = host_staged
        self.force_collective = force_collective  # tests: run the exchange even with one rank
        self.rank, self.world, self.device = rank, world, device
        self.min_shard = 64 * world if min_shard_batch is None else min_shard_batch
        self._buf = None
        self._buf2 = None
        self.n_collectives = 0
        # shard the deferred re-prunes of the apply phase too (needs the three-step apply of the C ABI)
        self.shard_apply = hasattr(builder, "apply_begin") if shard_apply is None else shard_apply
        self.min_shard_deferred = 32 * world if min_shard_deferred is None else min_shard_deferred
        # RCCL path: make the builder's stream torch's current stream around the collectives
        self._ext = None
        if (dist is not None and not host_staged and getattr(device, "type", "cpu") == "cuda"
                and getattr(builder, "stream_ptr", 0)):
            self._ext = torch.cuda.ExternalStream(builder.stream_ptr, device=device)

    def _on_builder_stream(self):
        return self.torch.cuda.stream(self._ext) if self._ext is not None else contextlib.nullcontext()

    def _buffer(self, words):
        if self._buf is None or self._buf.numel() < words:
            self.b.sync()  # the old buffer may still be in use on the builder's stream
            self._buf = self.torch.empty(words + words // 4, dtype=self.torch.int64, device=self.device)
        return self._buf

    @staticmethod
    def shard(count, world, rank):
        """members [lo, hi) of a batch searched by `rank`; equal-sized padded chunks"""
        per = math.ceil(count / world)
        lo = min(rank * per, count)
        return per, lo, min(lo + per, count)

    def run(self):
        b = self.b
        if self.dist is None or (self.world == 1 and not self.force_collective):
            return b.run()
        n = 0
        while True:
            bt = b.next_batch()
            if bt.count == 0:
                break
            n += 1
            if bt.count < self.min_shard:
                b.search(0, bt.count)
                b.apply()
                continue
            per, lo, hi = self.shard(bt.count, self.world, self.rank)
            stride = bt.sel_stride_u64
            full = self._buffer(self.world * per * stride)
            b.search(lo, hi, full.data_ptr())
            self._all_gather(full, per * stride)
            if not self.shard_apply:
                b.apply(full.data_ptr())
                continue
            nd = b.apply_begin(full.data_ptr())  # the same number on every rank (same inputs)
            if nd < self.min_shard_deferred:
                b.apply_deferred(0, 1, None)
                b.apply_merge(None, 0, 1)
                continue
            xs = b.exch_stride_u64
            per2 = math.ceil(nd / self.world)
            if self._buf2 is None or self._buf2.numel() < self.world * per2 * xs:
                b.sync()
                self._buf2 = self.torch.empty(self.world * per2 * xs * 2, dtype=self.torch.int64, device=self.device)
            ex = self._buf2
            b.apply_deferred(self.rank, self.world, ex.data_ptr())
            self._all_gather(ex, per2 * xs)
            b.apply_merge(ex.data_ptr(), self.rank, self.world)
        if getattr(b, "incremental", False):  # fill_gaps_from_deleted (hnsw.rs:187): deterministic,
            b.fill_gaps()                     # computed by every replica like Builder.run() does
        return n

    def _all_gather(self, full, words_per_rank):
        """in-place all-gather of `full[rank * w : (rank + 1) * w]` (w = words_per_rank), ordered
        after the builder's pending work and before its next step"""
        w = words_per_rank
        if self.host_staged or self._ext is None:
            self.b.sync()  # the builder runs on its own HIP stream
            mine = full[self.rank * w:(self.rank + 1) * w].clone()
            if self.host_staged:
                gathered = self.torch.empty(self.world * w, dtype=self.torch.int64)
                self.dist.all_gather_into_tensor(gathered, mine.cpu())
                full[:self.world * w].copy_(gathered)
            else:
                self.dist.all_gather_into_tensor(full[:self.world * w], mine)
            self.n_collectives += 1
            self._sync_collective()
            return
        try:
            with self._on_builder_stream():  # no host synchronisation: stream order does it
                mine = full[self.rank * w:(self.rank + 1) * w].clone()
                self.dist.all_gather_into_tensor(full[:self.world * w], mine)
        except (RuntimeError, TypeError, ValueError):  # a backend that refuses the external stream:
            self._ext = None                           # host-synchronised exchange from here on
            return self._all_gather(full, words_per_rank)
        self.n_collectives += 1

    def _sync_collective(self):
        if self.device is not None and getattr(self.device, "type", "cpu") == "cuda":
            self.torch.cuda.current_stream(self.device).synchronize()
