"""Item-sharded multi-GPU build driver: one process per GPU, every rank holds a full
replica of the vectors and of the graph in HBM.  Per batch each rank runs walk_layer + robust_prune
(hny_builder_search) for its contiguous slice of the batch members against the frozen graph, the
fixed-size selection records are exchanged with ONE all-gather (RCCL over xGMI, `nccl` backend),
and every rank applies the same link ops in the same order (hny_builder_apply), which keeps the
replicas bit-identical without a second collective.  Small batches (the ramp-up) are computed
redundantly by every rank instead: the exchange would cost more than the search.
"""
import math


class Driver:
    def __init__(self, builder, torch, dist, rank=0, world=1, device=None, min_shard_batch=None,
                 host_staged=False, force_collective=False):
        self.b, self.torch, self.dist = builder, torch, dist
        # host_staged: exchange through host memory (gloo); default is device buffers over RCCL
        self.host_staged = host_staged
        self.force_collective = force_collective  # tests: run the exchange even with one rank
        self.rank, self.world, self.device = rank, world, device
        self.min_shard = 64 * world if min_shard_batch is None else min_shard_batch
        self._buf = None
        self.n_collectives = 0

    def _buffer(self, words):
        if self._buf is None or self._buf.numel() < words:
            self._buf = self.torch.empty(words, dtype=self.torch.int64, device=self.device)
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
            b.sync()  # the builder runs on its own HIP stream
            mine = full[self.rank * per * stride:(self.rank + 1) * per * stride].clone()
            if self.host_staged:
                gathered = self.torch.empty(self.world * per * stride, dtype=self.torch.int64)
                self.dist.all_gather_into_tensor(gathered, mine.cpu())
                full[:self.world * per * stride].copy_(gathered)
            else:
                self.dist.all_gather_into_tensor(full[:self.world * per * stride], mine)
            self.n_collectives += 1
            self._sync_collective()
            b.apply(full.data_ptr())
        return n

    def _sync_collective(self):
        if self.device is not None and getattr(self.device, "type", "cpu") == "cuda":
            self.torch.cuda.current_stream(self.device).synchronize()
