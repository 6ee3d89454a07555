"""Database sharded by id range over the GPUs of one node: one process per GPU, per-shard top-k
exchanged with ONE all-gather (RCCL over xGMI via torch.distributed) and merged on every rank.

Replaces the reference's vertical partition + per-radius MPI_Gather/Gatherv/Bcast
(search_worker.cc:99-101,177,207; mpi_coordinator.cc:34-69): top-k of a union is the top-k of the per-shard
top-k's, so the only inter-GPU traffic is nq * k * 8 bytes per rank per batch, and no collective sits
inside the scan.

torch is plumbing here (device buffers, streams, the process group); all compute goes through the C ABI.
"""
import torch
import torch.distributed as dist

from . import engine as vc


def shard_range(total_n, rank, world):
    """ids [lo, hi) owned by `rank` (contiguous, balanced to within one item)."""
    return total_n * rank // world, total_n * (rank + 1) // world


class GpuBackend:
    """Local shard search + merge through libverticut_gpu.so."""

    def __init__(self, bits, capacity, id_base, n_tables=0, device=0, **kw):
        # the serving loop keeps its own clock: only the verify-kernel launches are bracketed by events
        kw["flags"] = kw.get("flags", 0) | vc.FLAG_LEAN_TIMING
        self.engine = vc.Engine(bits, capacity=max(capacity, 1), n_tables=n_tables, id_base=id_base, device=device, **kw)
        self.device = torch.device("cuda", device)

    def add_synthetic(self, n, seed, kind=vc.SYNTH_UNIFORM, n_centres=0, max_flips=0):
        self.engine.add_synthetic(n, seed, kind, n_centres, max_flips)

    def add_codes(self, codes):
        self.engine.add_codes(codes)

    def build_index(self):
        self.engine.build_index()

    def local_topk(self, queries, k, out, counts, mode):
        s = torch.cuda.current_stream(self.device).cuda_stream
        self.engine.search_knn_dev(queries.data_ptr(), queries.shape[0], k, out.data_ptr(), counts.data_ptr(), mode=mode,
                                   stream=s)

    def merge(self, gathered, world, nq, k, out, counts):
        s = torch.cuda.current_stream(self.device).cuda_stream
        vc.merge_topk_dev(gathered.data_ptr(), world, nq, k, out.data_ptr(), counts.data_ptr(), stream=s)

    def timing(self):
        return self.engine.timing()

    def get_code(self, gid):
        """ID -> BinaryCode of this shard (None when the id lives elsewhere)"""
        return self.engine.get_code(gid)

    def unrecovered(self):
        """calls whose device-side ring-overflow recovery gave up since the last query (vc_device_status); 0 normally"""
        return self.engine.device_status()

    def close(self):
        self.engine.close()


class PendingResult:
    """What a bucketed search() returns: the rows of one batch whose exchange happens with its bucket.

    get() -> (packed[nq, k] int64, counts[nq] int32), views into the bucket's buffers, enqueued-complete on the caller's
    stream; it exchanges the (partly filled) bucket first if that has not happened yet.  A bucket's buffers are reused by
    the bucket after the next one (two buffer sets alternate), so a handle must be consumed before two further buckets
    have been exchanged -- get() raises afterwards instead of returning recycled memory."""

    def __init__(self, owner, gen, lo, hi, bufs):
        self._owner, self._gen, self._lo, self._hi, self._bufs = owner, gen, lo, hi, bufs

    @property
    def ready(self):
        """the bucket's all-gather + merge have been enqueued (the tensors are valid in stream order)"""
        return self._owner._exchanged >= self._gen

    def get(self):
        o = self._owner
        if o._exchanged < self._gen:
            o.flush()
        if o._gen - self._gen >= 2 and o._fill_gen_started(self._gen + 2):
            raise RuntimeError("PendingResult consumed too late: its bucket buffers have been reused by a later bucket")
        out, ocnt = self._bufs
        return out[self._lo:self._hi], ocnt[self._lo:self._hi]

    def __iter__(self):            # `out, cnt = ss.search(...)` keeps working: unpacking waits for the bucket
        return iter(self.get())


class ShardedSearch:
    """k-NN over a database split across the ranks of `group` (default: WORLD).

    search(queries[nq, bits/8] uint8 on this rank's device, k) -> (packed[nq, k] int64 (bit pattern of the
    uint64 dist<<32|id, ascending, -1 = padding), counts[nq] int32), identical on every rank; with bucket > 1 a
    PendingResult handle whose get() returns that pair (it also unpacks like the pair).
    """

    def __init__(self, bits, total_n, rank=None, world=None, n_tables=0, device=None, group=None, backend=None,
                 pipelined=False, force_exchange=False, bucket=1, **engine_kw):
        """pipelined=True: the all-gather + merge of batch i run on a side stream while batch i+1 is already being
        scanned; results of a call are then ordered on the caller's stream only after flush() (or two calls later).
        bucket=B > 1: the per-shard top-k of B consecutive batches are exchanged with ONE all-gather + merge (the
        collective is latency-bound at 6.4 KB per rank, so B batches cost the same as one); search() then returns a
        PendingResult whose get() yields the tensors once the bucket has been exchanged (after B - 1 more calls, or at
        flush(), or on demand)."""
        self.group = group
        self.bucket = max(1, int(bucket))
        self._fill = 0
        self._bkey = None
        self._bsets = [None, None]     # two buffer sets alternate between bucket generations
        self._gen = 0                  # generation of the bucket being filled
        self._exchanged = -1           # last generation whose exchange has been enqueued
        self.pipelined = pipelined
        self.force_exchange = force_exchange   # run the all-gather + merge even with one rank (exercises RCCL on 1 GPU)
        self._side = None
        self._sets = [None, None]
        self._done = [None, None]
        self._step = 0
        self.world = world if world is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        self.rank = rank if rank is not None else (dist.get_rank(group) if dist.is_initialized() else 0)
        self.bits, self.nbytes, self.total_n = bits, bits // 8, total_n
        self.lo, self.hi = shard_range(total_n, self.rank, self.world)
        if total_n > 1 << 32:
            raise ValueError("ids are uint32 (image_search.proto:4): total_n must be <= 2^32")
        if backend is None:
            dev = device if device is not None else torch.cuda.current_device()
            backend = GpuBackend(bits, self.hi - self.lo, self.lo, n_tables=n_tables, device=dev, **engine_kw)
        self.backend = backend
        self._buf_key = None

    # -- data
    def add_synthetic(self, seed, kind=vc.SYNTH_UNIFORM, n_centres=0, max_flips=0):
        """every rank generates its own id range of the same global database"""
        self.backend.add_synthetic(self.hi - self.lo, seed, kind, n_centres, max_flips)

    def add_codes_global(self, codes):
        """codes = the whole database (row-major); this rank keeps rows [lo, hi)"""
        self.backend.add_codes(codes[self.lo:self.hi])

    def build_index(self):
        self.backend.build_index()

    # -- search
    def _buffers(self, nq, k, device):
        key = (nq, k, str(device))
        if self._buf_key != key:
            self._local = torch.empty((nq, k), dtype=torch.int64, device=device)
            self._lcnt = torch.empty((nq,), dtype=torch.int32, device=device)
            self._gath = torch.empty((self.world, nq, k), dtype=torch.int64, device=device)
            self._out = torch.empty((nq, k), dtype=torch.int64, device=device)
            self._ocnt = torch.empty((nq,), dtype=torch.int32, device=device)
            self._buf_key = key
        return self._local, self._lcnt, self._gath, self._out, self._ocnt

    def _search_pipelined(self, queries, k, mode):
        nq = queries.shape[0]
        j = self._step & 1
        self._step += 1
        key = (nq, k, str(queries.device))
        if self._sets[j] is None or self._sets[j][0] != key:
            dev = queries.device
            self._sets[j] = (key, torch.empty((nq, k), dtype=torch.int64, device=dev),
                             torch.empty((nq,), dtype=torch.int32, device=dev),
                             torch.empty((self.world, nq, k), dtype=torch.int64, device=dev),
                             torch.empty((nq, k), dtype=torch.int64, device=dev),
                             torch.empty((nq,), dtype=torch.int32, device=dev))
        _, local, lcnt, gath, out, ocnt = self._sets[j]
        main = torch.cuda.current_stream(queries.device)
        if self._side is None:
            self._side = torch.cuda.Stream(device=queries.device)
        if self._done[j] is not None:        # buffer set j was last used two batches ago: its exchange must be over
            main.wait_event(self._done[j])
        self.backend.local_topk(queries, k, local, lcnt, mode)
        ready = torch.cuda.Event()
        ready.record(main)
        with torch.cuda.stream(self._side):
            self._side.wait_event(ready)
            work = dist.all_gather_into_tensor(gath.view(self.world * nq, k), local, group=self.group, async_op=True)
            work.wait()                      # orders the side stream after the collective
            self.backend.merge(gath, self.world, nq, k, out, ocnt)
            done = torch.cuda.Event()
            done.record(self._side)
            self._done[j] = done
        return out, ocnt

    def flush(self):
        """complete what is pending: exchange a partly filled bucket, and make the caller's current stream wait for
        every side-stream exchange still in flight (pipelined mode)"""
        if self._fill:
            self._exchange_bucket()
        for ev in self._done:
            if ev is not None:
                torch.cuda.current_stream().wait_event(ev)

    def _fill_gen_started(self, gen):
        return self._gen > gen or (self._gen == gen and self._fill > 0)

    def _exchange_bucket(self):
        _, nq, k, local, lcnt, gath, out, ocnt = self._bkey
        rows = self._fill * nq
        self._fill = 0
        g = gath.view(-1, k)[: self.world * rows]                       # [world][rows][k], contiguous prefix
        dist.all_gather_into_tensor(g, local[:rows], group=self.group)   # rank-major concat
        self.backend.merge(g.view(self.world, rows, k), self.world, rows, k, out[:rows], ocnt[:rows])
        self._exchanged = self._gen
        self._gen += 1
        self._bkey = None                                                # the next bucket takes the other buffer set

    def _bucket_buffers(self, key, nq, k, dev):
        j = self._gen & 1
        if self._bsets[j] is None or self._bsets[j][0] != key:
            B = self.bucket
            self._bsets[j] = (key, nq, k, torch.empty((B * nq, k), dtype=torch.int64, device=dev),
                              torch.empty((B * nq,), dtype=torch.int32, device=dev),
                              torch.empty((self.world, B * nq, k), dtype=torch.int64, device=dev),
                              torch.empty((B * nq, k), dtype=torch.int64, device=dev),
                              torch.empty((B * nq,), dtype=torch.int32, device=dev))
        return self._bsets[j]

    def _search_bucketed(self, queries, k, mode):
        nq, dev, B = queries.shape[0], queries.device, self.bucket
        key = (nq, k, str(dev), mode)
        if self._bkey is not None and self._bkey[0] != key and self._fill:
            self._exchange_bucket()                                       # shape changed mid-bucket
        if self._bkey is None:
            self._bkey = self._bucket_buffers(key, nq, k, dev)
        _, _, _, local, lcnt, gath, out, ocnt = self._bkey
        lo, hi = self._fill * nq, (self._fill + 1) * nq
        self.backend.local_topk(queries, k, local[lo:hi], lcnt[lo:hi], mode)
        self._fill += 1
        res = PendingResult(self, self._gen, lo, hi, (out, ocnt))
        if self._fill == B:
            self._exchange_bucket()
        return res

    def search(self, queries, k, mode=vc.MODE_LINEAR):
        exchange = self.world > 1 or self.force_exchange
        if self.pipelined and exchange and queries.is_cuda:
            return self._search_pipelined(queries, k, mode)
        if self.bucket > 1 and exchange:
            return self._search_bucketed(queries, k, mode)
        nq = queries.shape[0]
        local, lcnt, gath, out, ocnt = self._buffers(nq, k, queries.device)
        self.backend.local_topk(queries, k, local, lcnt, mode)
        if self.world == 1 and not self.force_exchange:
            return local, lcnt
        # the only exchange step of the path: nq*k*8 bytes per rank
        dist.all_gather_into_tensor(gath.view(self.world * nq, k), local, group=self.group)  # rank-major concat
        self.backend.merge(gath, self.world, nq, k, out, ocnt)
        return out, ocnt

    def unrecovered(self):
        """Diagnostic of the asynchronous path: number of local search calls since the last check in which a
        candidate-ring overflow could NOT be recovered on the device (the affected rows are upper bounds only and the
        batch should be re-run through Engine.search_knn).  The device recovery is exact and self-contained, so this
        is 0 unless the GPU was held by foreign kernels for seconds.  Synchronises the local stream."""
        f = getattr(self.backend, "unrecovered", None)
        return f() if f else 0

    def close(self):
        self.backend.close()
