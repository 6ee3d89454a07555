#!/usr/bin/env python3
"""bench.py -- BASELINE.json headline: queries/sec, k-NN top-100, 128-bit codes, 1B-code database
(configs[2]; at --gpus N the same 1B database is sharded N ways = configs[3], strong scaling).

One step = one batch of Q queries verified against the whole database by the HBM-bound verify kernel
(vc_scan_kernel) + top-k select (+ all-gather/merge of per-shard top-k when N > 1).  Database, queries and
result buffers are resident in HBM before the timed region starts.

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` (achieved HBM GB/s of the
verify kernel from HIP events recorded on its stream, against the 8 TB/s peak; `traffic` = HBM bytes per launch from
a rocprofv3 --pmc pass of a short child run of this same script) and `cpu_baseline` (the CPU restatement of the
reference path timed on this host on a bounded sample).

    python bench.py                               # 1 GPU, 1e9 codes: the headline line
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
    python bench.py --workload c2                 # BASELINE configs[1]: 64-bit, 1e8 codes, MIH radius-8 search (m = 2; m = 4 as a variant)
    python bench.py --workload c5shard            # one GPU's share of configs[4]: 256-bit, 5e8 codes, 4096 queries per pass
    python bench.py --workload knn_mih            # exact top-100 through MIH on 1e8 clustered 128-bit codes (--approximate, --uniform-queries)
    python bench.py --workload c1                 # BASELINE configs[0]: 64-bit, 2^20 codes, 200 queries, linear (plumbing shape + CPU rows)
    python bench.py --workload sharded1dev        # 1e9 codes as 8 id-range shards on one device through vc_sharded_search_knn_dev
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

# MIH workloads: every N-th launch of the MIH kernels is bracketed by timing events (vc_config.timing_sample under VC_FLAG_LEAN_TIMING);
# an event record is a barrier packet of ~4 us in the stream, two of them per 0.3 ms launch are 3 % of a step
MIH_TIMING_SAMPLE = int(os.environ.get("VC_BENCH_MIH_TIMING_SAMPLE", "4"))
HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E peak (MI355X_MICROARCH.md; ~6.3 TB/s is what a pure copy reaches)
VALU_PEAK_GOPS = 39321.6  # 256 CUs x 64 lanes x 2.4 GHz 32-bit lane-ops/s; v_xor issues at ~2.3 and
#                           v_bcnt at ~4.2 cycles per wave64 instruction per SIMD, so the xor+popcount mix tops out near 0.6 of it
METRIC = "queries/sec (k-NN top-100) on 128-bit codes, 1B DB; bit-exact vs linear_search"


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c3", choices=["c3", "c1", "c2", "c5shard", "knn_mih", "sharded1dev"],
                    help="c3 = the headline (BASELINE configs[2]/[3]); the others are extra lines, same JSON shape")
    ap.add_argument("--db-size", dest="n", type=float, default=None, help="database size (total over all GPUs)")
    ap.add_argument("--bits", type=int, default=None)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--queries", type=int, default=None, help="queries per step")
    ap.add_argument("--seed", type=int, default=34, help="linear_search.cc:105 srand(34)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--tables", default="2,4", help="workload c2: table counts to run (the first one is the line's value)")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 --pmc child run that measures roofline.traffic")
    ap.add_argument("--uniform-queries", action="store_true", help="workload knn_mih: uniform random queries (the worst case of the radius "
                    "loop: shells up to r ~ 8; answered through the cost-model switch to the verify kernel)")
    ap.add_argument("--no-extras", action="store_true", help="default line only: skip the short measurements of the other shapes (`extras`)")
    ap.add_argument("--approximate", action="store_true", help="workload knn_mih: VC_MODE_MIH_APPROX (search_worker.cc:93-157) instead of the exact loop")
    ap.add_argument("--shards", type=int, default=8, help="workload sharded1dev: id-range shards behind vc_sharded_* (all on device 0)")
    args = ap.parse_args(argv)
    d = {"c3": (1e9, 128, 8, 30), "c1": (1 << 20, 64, 200, 50), "c2": (1e8, 64, 1024, 20), "c5shard": (5e8, 256, 4096, 4),
         "knn_mih": (1e8, 128, 4096, 10), "sharded1dev": (1e9, 128, 8, 30)}[args.workload]
    if args.n is None:
        args.n = d[0]
    if args.bits is None:
        args.bits = d[1]
    if args.queries is None:
        args.queries = d[2]
    if args.steps is None:
        args.steps = d[3]
    return args


# ---------------------------------------------------------------------------------------------------------------
# CPU baselines (the only place bench.py touches oracle/, outside every timed GPU region)
# ---------------------------------------------------------------------------------------------------------------
def cpu_baseline_linear(args, n_total):
    """linear_search.cc:39-64 restated (oracle/vc_oracle.cc vco_linear_knn_ref: 32-bit popcount loop +
    std::priority_queue), 1 thread, in-memory codes, on a bounded sample of the same synthetic database;
    scaled to the full database size (the scan is exactly linear in N)."""
    from oracle import vc_oracle as vo
    sample_n = int(min(n_total, 1 << 24))
    codes = vo.gen_codes(sample_n, args.bits, args.seed)
    rng = np.random.default_rng(12345)
    q = rng.integers(0, 256, size=(4096, args.bits // 8), dtype=np.uint8)
    vo.linear_knn_ref(codes[:100000], q[0], args.k)  # warm
    t0 = time.perf_counter()
    done = 0
    while done < len(q) and time.perf_counter() - t0 < args.cpu_seconds:   # bounded: ~cpu_seconds of CPU work
        vo.linear_knn_ref(codes, q[done], args.k)
        done += 1
    dt = time.perf_counter() - t0
    qps_sample = done / dt
    nproc = os.cpu_count() or 1
    res = {
        "value": qps_sample * sample_n / n_total,
        "unit": "queries/s",
        "cores": 1,
        "nproc": nproc,
        "kind": "port",
        "sample": "%d queries x first %d codes of the same synthetic DB in %.1f s, 1 thread; scaled by N_sample/N "
                  "(scan cost is linear in N); structure of linear_search.cc:39-64 over in-memory codes (no KV get)"
                  % (done, sample_n, dt),
        "items_per_s": qps_sample * sample_n,
    }
    # every host core this job may use (BASELINE.md row CPU-linear-allcores): the same scan on a persistent worker pool
    # (oracle/vc_oracle.cc VcoPool: threads created once, every worker scans its id range for the whole query batch,
    # per-range heaps merged per query), in batches of the GPU step's size
    with vo.Pool() as pool:
        # a bigger sample than the 1-thread leg's: every worker should stream megabytes per call, not kilobytes
        big_n = int(min(n_total, 1 << 27))
        big = pool.gen_codes(big_n, args.bits, args.seed) if big_n > sample_n else codes
        batch = max(1, min(args.queries, 64))
        pool.linear_knn(big[:1 << 20], q[:batch], args.k)  # warm
        t0 = time.perf_counter()
        done = 0
        while done + batch <= len(q) and time.perf_counter() - t0 < max(2.0, args.cpu_seconds / 3):
            pool.linear_knn(big, q[done:done + batch], args.k)
            done += batch
        dt = time.perf_counter() - t0
        res["allcores"] = {"value": done / dt * big_n / n_total, "cores": pool.threads, "nproc": nproc,
                           "items_per_s": done / dt * big_n,
                           "sample": "%d queries in batches of %d x first %d codes in %.1f s on a persistent pool of %d threads; scaled by N_sample/N"
                                     % (done, batch, big_n, dt, pool.threads)}
    return res


def cpu_baseline_mih(args, m, kind, radius=None, clustered=False, approximate=False):
    """search_worker.cc:159-264 restated (oracle/vc_oracle.cc: enumerate_entry per rank, gather in rank order, master-side
    dedup + heap), one thread per table like `mpirun -n m` (run_distributed_search.py:12,74), in-memory buckets instead of
    a KV tier, on a bounded SAMPLE database: the probe count per query does not depend on N, only the bucket sizes do,
    so the sample flatters the CPU (fewer candidates per bucket than at full size) and is reported unscaled."""
    from oracle import vc_oracle as vo
    sample_n = int(min(args.n, 2_000_000 if clustered else 4_000_000))
    t0 = time.perf_counter()
    if clustered:
        codes = vo.gen_codes(sample_n, args.bits, args.seed, kind=1, n_centres=max(sample_n // 1000, 1), max_flips=11)
    else:
        codes = vo.gen_codes(sample_n, args.bits, args.seed)
    mo = vo.MihOracle(codes, m, key_mode=1)
    t_build = time.perf_counter() - t0
    rng = np.random.default_rng(4321)
    nproc = os.cpu_count() or 1
    threads = min(m, nproc)
    t0 = time.perf_counter()
    done, probes = 0, 0
    while time.perf_counter() - t0 < args.cpu_seconds:
        q = codes[int(rng.integers(0, sample_n))].copy()
        for b in rng.choice(args.bits, size=int(rng.integers(0, (radius if radius is not None else 4) + 1)), replace=False):
            q[b // 8] ^= np.uint8(1 << (b % 8))
        if kind == "radius":
            _, pr = mo.radius(q, radius, threads=threads)
            probes += pr
        else:
            _, st = mo.find(q, args.k, stop_mult=min(m, 4), threads=threads, approximate=approximate)
            probes += st.n_sub_reads_all
        done += 1
    dt = time.perf_counter() - t0
    return {
        "value": done / dt, "unit": "queries/s", "cores": threads, "nproc": nproc, "kind": "port",
        "sample": "%d queries in %.1f s on a %d-code sample of the same synthetic DB (index build %.1f s not counted), %d threads "
                  "= one per table; %s; NOT scaled to the full size (bucket sizes grow with N, probes per query do not: %.0f "
                  "bucket gets per query)" % (done, dt, sample_n, t_build, threads,
                                              "search_R_neighbors shells 0..q per table, q = r/m for the first r mod m + 1 tables and r/m - 1 for the rest, + gather + dedup (search_worker.cc:222-264)"
                                              if kind == "radius" else ("SearchWorker::find approximate loop (search_worker.cc:93-157)" if approximate
                                                                        else "SearchWorker::find exact loop (search_worker.cc:159-218)"),
                                              probes / max(done, 1)),
    }


# ---------------------------------------------------------------------------------------------------------------
# environment: ranks, devices, process group
# ---------------------------------------------------------------------------------------------------------------
class Env:
    """rank / world / device of this process and the small set of collectives the bench needs"""

    def __init__(self, args, device_kind="cuda", dist_backend=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus:
            if self.world == 1 and args.gpus > 1:
                raise SystemExit("--gpus %d needs torch.distributed.run with %d ranks" % (args.gpus, args.gpus))
            args.gpus = self.world
        # rehearsal knobs (dev only): run the multi-rank flow on ONE GPU, where RCCL cannot be used (one device per rank)
        self.backend = dist_backend or os.environ.get("VC_BENCH_BACKEND", "nccl")
        if os.environ.get("VC_BENCH_ONE_GPU") == "1":
            self.local_rank = 0
        self.cuda = device_kind == "cuda"
        if self.cuda:
            torch.cuda.set_device(self.local_rank)
            self.device = torch.device("cuda", self.local_rank)
        else:
            self.device = torch.device("cpu")
        self.own_group = False
        if self.world > 1 and not dist.is_initialized():
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            if self.backend == "nccl":
                dist.init_process_group("nccl", device_id=self.device)
            else:
                dist.init_process_group(self.backend)
            self.own_group = True

    def sync(self):
        if self.cuda:
            self.torch.cuda.synchronize()

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()

    def max_over_ranks(self, x):
        if self.world == 1:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def all_ok(self, ok):
        if self.world == 1:
            return ok
        t = self.torch.tensor([1 if ok else 0], device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return bool(t.item())

    def close(self):
        if self.own_group:
            self.dist.destroy_process_group()


def timed_steps(env, run_steps, steps):
    """the contract's timed region: barrier + synchronize on both sides, MAX over ranks"""
    env.barrier()
    env.sync()
    t0 = time.perf_counter()
    res = run_steps(steps)
    env.sync()
    t1 = time.perf_counter()
    env.barrier()
    return res, env.max_over_ranks(t1 - t0)


def step_times_ms(env, run_one, steps):
    """Per-step device times (SURVEY.md 8d asks for a median next to the mean): an event between consecutive steps on the
    launch stream, in a pass of its own AFTER the contract's timed region (whose clock stays two host timestamps around
    K steps).  run_one(i) enqueues step i on torch's current stream."""
    torch = env.torch
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    env.sync()
    ev[0].record()
    for i in range(steps):
        run_one(i)
        ev[i + 1].record()
    env.sync()
    return [ev[i].elapsed_time(ev[i + 1]) for i in range(steps)]


def _median(v):
    return float(np.median(np.asarray(v, dtype=np.float64))) if len(v) else None


def measure_traffic(args, kernel_substr, fetch_mult=2.0, extra=()):
    """roofline.traffic: HBM bytes per launch of the dominant kernel from the PMC counters of a short child run of this
    same script under rocprofv3 (separate --pmc passes for FETCH_SIZE and WRITE_SIZE, TCC slots; FETCH_SIZE x 2 for a
    wide coalesced stream on gfx950 -- MI355X_MICROARCH.md, HBM section).  None when rocprofv3 is unavailable or fails."""
    prof = shutil.which("rocprofv3")
    if prof is None:
        return None, "rocprofv3 not on PATH"
    import csv
    import glob
    vals = {}
    base = ["--workload", args.workload, "--db-size", repr(float(args.n)), "--bits", str(args.bits), "--k", str(args.k),
            "--queries", str(args.queries), "--seed", str(args.seed), "--steps", "3", "--warmup", "1", "--cpu-seconds", "0",
            "--no-check", "--no-traffic", "--no-extras"] + list(extra)
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        with tempfile.TemporaryDirectory(dir="/tmp") as td:
            env = dict(os.environ, TMPDIR="/tmp")
            cmd = [prof, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", td, "--",
                   sys.executable, os.path.join(ROOT, "bench.py")] + base
            # the child gets its own process group so that a hung profiler run can be ended as a whole (exact pgid)
            try:
                proc = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            except OSError as ex:
                return None, "rocprofv3 could not be started: %s" % ex
            try:
                rc = proc.wait(timeout=150)
            except subprocess.TimeoutExpired:
                import signal
                try:
                    os.killpg(proc.pid, signal.SIGKILL)
                except OSError:
                    pass
                proc.wait()
                return None, "rocprofv3 --pmc %s child run timed out" % counter
            if rc != 0:
                return None, "rocprofv3 --pmc %s child run failed (exit %d)" % (counter, rc)
            got = []
            for f in glob.glob(os.path.join(td, "**", "*_counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        if kernel_substr in row["Kernel_Name"] and row["Counter_Name"] == counter:
                            got.append(float(row["Counter_Value"]))
            if not got:
                return None, "no %s rows for %s" % (counter, kernel_substr)
            vals[counter] = sum(got[1:]) / max(len(got) - 1, 1) if len(got) > 1 else got[0]   # drop the first (cold) launch
    return (vals["FETCH_SIZE"] * fetch_mult + vals["WRITE_SIZE"]) * 1024.0, \
        "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, KB) of a 3-step child run: FETCH_SIZE x %g + WRITE_SIZE" % fetch_mult


# ---------------------------------------------------------------------------------------------------------------
# headline: BASELINE configs[2] (1 GPU) / configs[3] (N GPUs)
# ---------------------------------------------------------------------------------------------------------------
def run_headline(args, env, emit, backend_factory=None):
    torch = env.torch
    from verticut_amd.sharded import ShardedSearch
    world, rank = env.world, env.rank
    n_total = int(args.n)
    Q, k = args.queries, args.k
    force_exchange = world == 1 and os.environ.get("VC_BENCH_FORCE_EXCHANGE") == "1"   # dev: RCCL exchange with one rank
    if force_exchange and not env.dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        env.dist.init_process_group(env.backend, rank=0, world_size=1, **({"device_id": env.device} if env.backend == "nccl" else {}))
        env.own_group = True
    engine_kw = {}
    if world > 1:        # shards are small (0.35 ms per pass): time every 8th verify launch only, an event record costs ~5 us
        engine_kw["timing_sample"] = int(os.environ.get("VC_BENCH_TIMING_SAMPLE", "8"))
    elif os.environ.get("VC_BENCH_TIMING_SAMPLE"):
        engine_kw["timing_sample"] = int(os.environ["VC_BENCH_TIMING_SAMPLE"])
    if os.environ.get("VC_BENCH_SCAN_BLOCKS"):      # dev: cap the persistent verify grid (leave block slots to other kernels)
        engine_kw["scan_blocks"] = int(os.environ["VC_BENCH_SCAN_BLOCKS"])
    # N > 1: the per-shard top-k of 8 consecutive batches share one all-gather + merge (ShardedSearch(bucket=8)): the
    # collective is latency-bound at 6.4 KB per rank, every step still ends inside the timed region (flush()).
    bucket = int(os.environ.get("VC_BENCH_BUCKET", "8")) if (world > 1 or force_exchange) else 1
    if backend_factory is not None:
        from verticut_amd.sharded import shard_range
        lo, hi = shard_range(n_total, rank, world)
        ss = ShardedSearch(args.bits, n_total, rank=rank, world=world, backend=backend_factory(args.bits, lo, hi),
                           force_exchange=force_exchange, bucket=bucket)
    else:
        qtile = int(os.environ.get("VC_BENCH_QUERY_TILE", Q))   # dev: several verify launches per step (tools/multi_tile.sh)
        ss = ShardedSearch(args.bits, n_total, rank=rank, world=world, device=env.local_rank, query_tile=qtile,
                           force_exchange=force_exchange, bucket=bucket, **engine_kw)
    ss.add_synthetic(args.seed)

    # query batches resident in HBM: uniform random codes = worst case (no early threshold help)
    rng = np.random.default_rng(args.seed + 1)
    nb = 4
    host_q = [rng.integers(0, 256, size=(Q, args.bits // 8), dtype=np.uint8) for _ in range(nb)]
    dev_q = [torch.from_numpy(h).to(env.device) for h in host_q]
    env.sync()

    def run_steps(count):
        res = None
        for i in range(count):
            res = ss.search(dev_q[i % nb], k)
        ss.flush()                  # N > 1: the last (partly filled) bucket is exchanged inside the timed region
        return res.get() if hasattr(res, "get") else res    # bucketed exchange hands out PendingResult handles

    # Priming, part of set-up like the data generation above: the first launches after the 16 GB fill run 10-15 %
    # slow (clocks, page tables; kernel trace in profiles/), and a driver-chosen --warmup may be shorter than that.
    run_steps(8)
    env.sync()
    run_steps(args.warmup)
    env.sync()
    exchange = "none"
    if world > 1 or force_exchange:
        # The per-batch exchange (all-gather + merge) runs inline on the step's stream: the plain, widely used pattern.
        # VC_BENCH_PIPELINED=1 moves it to a side stream under the next batch's scan (ShardedSearch(pipelined=True));
        # that path is covered by tests but has never run over RCCL on a multi-GPU node, so it is opt-in.
        ss.pipelined = os.environ.get("VC_BENCH_PIPELINED") == "1" and env.cuda
        exchange = "side-stream" if ss.pipelined else ("inline, %d batches per all-gather" % bucket if bucket > 1 else "inline")
        if ss.pipelined:
            run_steps(2)
            env.sync()
    ss.backend.timing()  # drop warm-up event records

    (out, cnt), elapsed = timed_steps(env, run_steps, args.steps)
    tm = ss.backend.timing()  # HIP events on the launch stream, exactly the timed steps
    out, cnt = out.clone(), cnt.clone()   # the passes below reuse the result buffers
    step_ms = None
    if env.cuda and backend_factory is None:
        def one(i):
            r = ss.search(dev_q[i % nb], k)
            if (i + 1) % bucket == 0 or i + 1 == args.steps:
                ss.flush()
            return r
        step_ms = step_times_ms(env, one, args.steps)
        ss.backend.timing()
    pre_extras = {}
    if world == 1 and env.cuda and backend_factory is None and not args.no_extras and not force_exchange:
        for name, fn in (("e2e_host", lambda: _extra_e2e_host(args, env, ss.backend.engine, host_q)),
                         ("q1", lambda: _extra_q1(args, env, ss.backend.engine, dev_q))):
            try:
                pre_extras[name] = fn()
            except Exception as ex:
                pre_extras[name] = {"error": "%s: %s" % (type(ex).__name__, ex)}

    ok = True
    if not args.no_check:
        # size-independent properties of the last batch (bit-exact parity proper lives in tests/): ascending
        # packed values, k results, every reported distance recomputed from the stored code, no unrecovered overflow.
        res = out.cpu().numpy().view(np.uint64)
        qh = host_q[(args.steps - 1) % nb]
        why = []
        if not np.all(cnt.cpu().numpy() == min(k, n_total)):
            why.append("counts %s" % cnt.cpu().numpy().tolist())
        if not np.all(res[:, 1:] > res[:, :-1]):
            why.append("rows not strictly ascending")
        for qi in range(min(Q, 2)):
            for j in range(k):                            # every result of two queries that this rank's shard holds
                gid = int(res[qi, j] & np.uint64(0xFFFFFFFF))
                if ss.lo <= gid < ss.hi:
                    code = ss.backend.get_code(gid)
                    d = int(np.unpackbits(np.bitwise_xor(code, qh[qi])).sum())
                    if d != int(res[qi, j] >> np.uint64(32)):
                        why.append("query %d result %d: id %d reported %d, stored code says %d"
                                   % (qi, j, gid, int(res[qi, j] >> np.uint64(32)), d))
        if ss.unrecovered():
            why.append("device-side ring-overflow recovery gave up")
        ok = not why
        if why:
            sys.stderr.write("[bench rank %d] results check failed: %s\n" % (rank, "; ".join(why[:4])))
        ok = env.all_ok(ok)
    ss.close()

    if rank == 0:
        scan_avg_ms = tm.scan_ms / max(tm.scan_launches, 1)
        bytes_per_launch = tm.scan_bytes / max(tm.scan_launches, 1)
        resident_mb = int(os.environ.get("VC_SCAN_RESIDENT_MB", "240"))          # VC_RESIDENT_MB_DEFAULT (vc_engine.hip)
        resident_bytes = min(resident_mb << 20, int(bytes_per_launch))
        achieved = bytes_per_launch / (scan_avg_ms * 1e-3) / 1e9 if scan_avg_ms > 0 else 0.0
        traffic, traffic_how = None, "not measured"
        if world == 1 and env.cuda and not args.no_traffic and backend_factory is None:
            traffic, traffic_how = measure_traffic(args, "vc_scan_kernel")
        line = {
            "metric": METRIC,
            "value": Q * args.steps / elapsed,
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "median_ms_per_step": _median(step_ms) if step_ms else None,   # per-step events, a pass of its own after the timed region
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[%d]: %d-bit codes, %.3g-code DB%s, top-%d k-NN, linear verify kernel"
                            % (2 if world == 1 else 3, args.bits, n_total,
                               "" if world == 1 else " sharded %d ways by id range" % world, k),
                "n_codes": n_total, "bits": args.bits, "k": k, "queries_per_step": Q,
                "query_tile": int(os.environ.get("VC_BENCH_QUERY_TILE", Q)),
                "query_kind": "uniform random", "seed": args.seed,
                "parallelism": "1 process/GPU, DB shard per GPU, RCCL all-gather of per-shard top-k + merge kernel"
                               if world > 1 else "single GPU",
                "exchange": exchange,
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_how": traffic_how,
                "kernel": "vc_scan_kernel", "launches": tm.scan_launches, "avg_launch_ms": scan_avg_ms,
                "algorithmic_bytes_per_launch": bytes_per_launch,
                # `achieved` is ALGORITHMIC bandwidth (every code byte once per launch).  The first resident_prefix_bytes of
                # the database are read with cache-allocating loads and stay in the 256 MB Infinity Cache between passes
                # (DESIGN.md 4.1): from the second pass on they do not come from HBM -- 1.5 % of a 16 GB database, 12 % of a
                # 2 GB shard -- and FETCH_SIZE counts them all the same.  achieved_excl_resident prices only the rest.
                "resident_prefix_bytes": resident_bytes,
                "achieved_excl_resident": (max(bytes_per_launch - resident_bytes, 0) / (scan_avg_ms * 1e-3) / 1e9) if scan_avg_ms > 0 else 0.0,
            },
            "results_check": "ok" if ok else "FAILED",
        }
        if world == 1 and args.cpu_seconds > 0:
            line["cpu_baseline"] = cpu_baseline_linear(args, n_total)
        if pre_extras:
            line["extras"] = dict(pre_extras)
        if world == 1 and env.cuda and backend_factory is None and not args.no_extras and n_total >= 10 ** 9:
            line.setdefault("extras", {}).update(run_extras(args, env))
        emit(line)
    return ok


# ---------------------------------------------------------------------------------------------------------------
# extra workloads (1 GPU): same JSON shape, never the default
# ---------------------------------------------------------------------------------------------------------------
def _near_queries(engine, n, nq, bits, max_flips, rng):
    q = np.empty((nq, bits // 8), dtype=np.uint8)
    for i in range(nq):
        c = engine.get_code(int(rng.integers(0, n)))
        for b in rng.choice(bits, size=int(rng.integers(0, max_flips + 1)), replace=False):
            c[b // 8] ^= np.uint8(1 << (b % 8))
        q[i] = c
    return q


def _mih_roofline(tm, bits, kernel="mih_query_kernel"):
    """bytes = probes x 4 (bitmap) + non-empty buckets x 16 (key lookup) + entries x (4 id + B/8 code)"""
    if tm.mih_launches == 0:
        return {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": None, "traffic": None,
                "kernel": "mih_probe_kernel", "note": "this shape (buckets of ~1500 entries) runs the multi-block shell kernels, "
                "one launch sequence per shell; only mih_query_kernel is instrumented"}
    launches = max(tm.mih_launches, 1)
    if kernel == "mih_bucket_stream_kernel":
        # the streaming kernel reads two offsets per probe and the CODES of the bucket entries (bucket-order copy); ids are
        # gathered for results only -- SURVEY 8(d)'s per-entry figure (id + code) would price bytes nobody reads
        alg = (tm.mih_probes * 8 + tm.mih_entries * (bits // 8)) / launches
    else:
        alg = (tm.mih_probes * 4 + tm.mih_hits * 16 + tm.mih_entries * (4 + bits // 8)) / launches
    avg_ms = tm.mih_ms / launches
    achieved = alg / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    return {
        "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
        "traffic": None, "kernel": kernel, "launches": tm.mih_launches, "avg_launch_ms": avg_ms,
        "algorithmic_bytes_per_launch": alg,
        "per_query": {"probes": tm.mih_probes / max(tm.mih_queries, 1), "non_empty_buckets": tm.mih_hits / max(tm.mih_queries, 1),
                      "entries_verified": tm.mih_entries / max(tm.mih_queries, 1)},
        "note": "random 4..64-byte accesses: the byte roofline is the wrong yardstick by construction (sector-granular "
                "gathers); sector traffic from rocprofv3 is in profiles/ and DESIGN.md 4.2",
    }


# The yardstick that fits the MIH kernels: 64-byte requests per second against the random-sector ceiling of the memory
# system, measured with tools/ubench_sectors.hip (independent 4..64-byte loads from uniformly random 64-byte sectors,
# profiles/r03_ubench_sectors.txt): 54 G sectors/s over a 2 GB footprint (the four occupancy bitmaps), 49-50 G over
# 16-128 GB (the bucket-order records at 1e9).
SECTOR_PEAK_G = 54.0


def _add_sector_roofline(roof):
    """roofline.sectors: the kernel's measured 64-byte requests per launch / its launch time, against the random-sector ceiling"""
    if roof.get("traffic") and roof.get("avg_launch_ms"):
        ach = roof["traffic"] / 64.0 / (roof["avg_launch_ms"] * 1e-3) / 1e9
        roof["sectors"] = {"achieved": ach, "peak": SECTOR_PEAK_G, "unit": "G 64-byte requests/s", "frac": ach / SECTOR_PEAK_G,
                           "peak_how": "tools/ubench_sectors.hip on one MI355X: independent random-sector loads, 2 GB footprint "
                                       "(profiles/r03_ubench_sectors.txt; 49-50 G over 16-128 GB)"}


# ---------------------------------------------------------------------------------------------------------------
# extras of the default line: short measurements of the other shapes in the SAME invocation, so that the driver's
# record carries them too (each: whole-call queries/s, the dominant kernel's average launch time and its algorithmic
# bytes per launch; the full lines with cpu_baseline and PMC traffic are `--workload c2 / knn_mih`)
# ---------------------------------------------------------------------------------------------------------------
def _scan_fields(tm, steps=None):
    """kernel time, algorithmic bytes and fraction of the HBM peak of the verify-kernel launches in `tm`"""
    launches = max(tm.scan_launches, 1)
    avg = tm.scan_ms / launches
    alg = tm.scan_bytes / launches
    ach = alg / (avg * 1e-3) / 1e9 if avg > 0 else 0.0
    return {"kernel": "vc_scan_kernel", "kernel_avg_ms": avg, "kernel_launches": tm.scan_launches, "algorithmic_bytes_per_launch": alg,
            "achieved_GBps": ach, "frac": ach / HBM_PEAK_GBPS}


def _extra_e2e_host(args, env, e, host_q, steps=12):
    """What a SearchWorker::find caller sees (distributed_image_search.cc:62-85): the same 8-query step through
    vc_search_knn with HOST pointers -- the queries' H2D copy and the rows' / counts' D2H copy inside the timed region,
    every call synchronous (SURVEY.md 8d "end-to-end QPS including H2D of queries and D2H of Q x k x 8 B")."""
    Q, k = host_q[0].shape[0], args.k
    for i in range(3):
        e.search_knn(host_q[i % len(host_q)], k)
    e.timing()
    times = []
    t0 = time.perf_counter()
    for i in range(steps):
        t = time.perf_counter()
        rows, cnt = e.search_knn(host_q[i % len(host_q)], k)
        times.append((time.perf_counter() - t) * 1e3)
    elapsed = time.perf_counter() - t0
    tm = e.timing()
    ok = bool(np.all(cnt == k)) and bool(np.all(rows[:, 1:] > rows[:, :-1]))
    r = {"workload": "configs[2] step through vc_search_knn with host pointers: H2D of %d queries + D2H of %d x %d x 8 B rows inside the timed region, synchronous calls"
                     % (Q, Q, k), "value": Q * steps / elapsed, "unit": "queries/s", "ms_per_step": elapsed / steps * 1e3,
         "median_ms_per_step": _median(times), "results_check": "ok" if ok else "FAILED"}
    r.update(_scan_fields(tm))
    r["step_frac"] = r["algorithmic_bytes_per_launch"] / (elapsed / steps) / 1e9 / HBM_PEAK_GBPS   # bytes over the WHOLE step, PCIe included
    return r


def _extra_q1(args, env, e, dev_q, steps=12):
    """SURVEY.md 8d "C3 at Q = 1": one query per database pass (the pure HBM-bound form of the verify kernel), device API"""
    torch = env.torch
    k = args.k
    d_out = torch.empty((1, k), dtype=torch.int64, device=env.device)
    d_cnt = torch.empty((1,), dtype=torch.int32, device=env.device)
    st = torch.cuda.current_stream().cuda_stream
    q1 = [dq[j:j + 1].contiguous() for dq in dev_q for j in (0, 1)]

    def one(i):
        e.search_knn_dev(q1[i % len(q1)].data_ptr(), 1, k, d_out.data_ptr(), d_cnt.data_ptr(), stream=st)

    for i in range(3):
        one(i)
    env.sync()
    e.timing()
    _, elapsed = timed_steps(env, lambda c: [one(i) for i in range(c)], steps)
    tm = e.timing()
    res = d_out.cpu().numpy().view(np.uint64)
    ok = bool(np.all(d_cnt.cpu().numpy() == k)) and bool(np.all(res[:, 1:] > res[:, :-1]))
    times = step_times_ms(env, one, steps)
    e.timing()
    r = {"workload": "configs[2] at ONE query per pass (vc_search_knn_dev, nq = 1)", "value": steps / elapsed, "unit": "queries/s",
         "ms_per_step": elapsed / steps * 1e3, "median_ms_per_step": _median(times), "results_check": "ok" if ok else "FAILED"}
    r.update(_scan_fields(tm))
    return r


def _extra_qt32(args, env, steps=6):
    """the same verify kernel at 32 queries per pass: the throughput optimum (VALU-bound), where 8 per pass is the HBM-bound one"""
    torch = env.torch
    from verticut_amd import engine as vc
    n, Q, k = int(args.n), 32, args.k
    rng = np.random.default_rng(args.seed + 7)
    with vc.Engine(args.bits, capacity=n, query_tile=Q, flags=vc.FLAG_LEAN_TIMING) as e:
        e.add_synthetic(n, seed=args.seed)
        dq = [torch.from_numpy(rng.integers(0, 256, size=(Q, args.bits // 8), dtype=np.uint8)).to(env.device) for _ in range(2)]
        d_out = torch.empty((Q, k), dtype=torch.int64, device=env.device)
        d_cnt = torch.empty((Q,), dtype=torch.int32, device=env.device)
        st = torch.cuda.current_stream().cuda_stream

        def run_steps(count):
            for i in range(count):
                e.search_knn_dev(dq[i % 2].data_ptr(), Q, k, d_out.data_ptr(), d_cnt.data_ptr(), stream=st)

        run_steps(4)
        env.sync()
        e.timing()
        _, elapsed = timed_steps(env, run_steps, steps)
        tm = e.timing()
        res = d_out.cpu().numpy().view(np.uint64)
        ok = bool(np.all(d_cnt.cpu().numpy() == k)) and bool(np.all(res[:, 1:] > res[:, :-1])) and e.device_status() == 0
    r = {"workload": "configs[2] at 32 queries per pass (same engine and kernel; VALU-bound)", "value": Q * steps / elapsed,
         "unit": "queries/s", "ms_per_step": elapsed / steps * 1e3, "results_check": "ok" if ok else "FAILED"}
    r.update(_scan_fields(tm))
    return r


def _mih_fields(tm, bits, steps, kernel="mih_query_kernel"):
    """time, algorithmic bytes and fraction of the HBM peak of the MIH query kernel's launches in `tm`"""
    roof = _mih_roofline(tm, bits, kernel)
    return {"kernel": kernel, "kernel_ms_per_step": tm.mih_ms / steps, "kernel_launches_per_step": tm.mih_launches / steps,
            "kernel_avg_ms": roof.get("avg_launch_ms"),
            "algorithmic_bytes_per_step": (roof.get("algorithmic_bytes_per_launch") or 0) * tm.mih_launches / steps,
            "algorithmic_bytes_per_launch": roof.get("algorithmic_bytes_per_launch"),
            "achieved_GBps": roof.get("achieved"), "frac": roof.get("frac"), "per_query": roof.get("per_query")}


def _extra_c2(args, env, steps=6):
    """BASELINE configs[1] (64-bit, 1e8, all within 8, m = 2) -- the shape of `--workload c2`, a few steps"""
    torch = env.torch
    from verticut_amd import engine as vc
    n, bits, m, Q, radius = 100_000_000, 64, 2, 1024, 8
    rng = np.random.default_rng(args.seed + 2)
    e = vc.Engine(bits, capacity=n, n_tables=m, flags=vc.FLAG_LEAN_TIMING, timing_sample=MIH_TIMING_SAMPLE)
    try:
        e.add_synthetic(n, seed=args.seed)
        e.build_index()
        host_q = [_near_queries(e, n, Q, bits, 8, rng) for _ in range(2)]
        dq = [torch.from_numpy(h).to(env.device) for h in host_q]
        st = torch.cuda.current_stream().cuda_stream
        out_cap = Q * 64
        d_out = torch.empty((out_cap,), dtype=torch.int64, device=env.device)
        d_off = torch.empty((Q + 1,), dtype=torch.int64, device=env.device)

        def one(i):
            if e.search_radius_dev(dq[i % 2].data_ptr(), Q, radius, d_out.data_ptr(), out_cap, d_off.data_ptr(),
                                   mode=vc.MODE_MIH_EXACT, stream=st) != vc.VC_OK:
                raise SystemExit("extras c2: results do not fit")

        for i in range(3):
            one(i)
        env.sync()
        e.timing()
        _, elapsed = timed_steps(env, lambda c: [one(i) for i in range(c)], steps)
        tm = e.timing()
        qh = host_q[(steps - 1) % 2][:16]    # MIH == full scan on the last batch
        off = d_off.cpu().numpy().view(np.uint64)
        res = d_out.cpu().numpy().view(np.uint64)
        lin = e.search_radius(qh, radius, mode=vc.MODE_LINEAR)
        ok = all(np.array_equal(res[int(off[i]):int(off[i + 1])], lin[i]) for i in range(16))
        times = step_times_ms(env, one, steps)
    finally:
        e.close()
    r = {"workload": "configs[1]: 64-bit, 1e8 codes, all neighbours within 8, MIH m=2, 1024 queries per call",
         "value": Q * steps / elapsed, "unit": "queries/s", "ms_per_step": elapsed / steps * 1e3, "median_ms_per_step": _median(times),
         "results_check": "ok" if ok else "FAILED"}
    r.update(_mih_fields(tm, bits, steps))
    return r


def _extras_knn_mih(args, env, n, tag, legs, steps=6):
    """SearchWorker::find through MIH on `n` clustered 128-bit codes (n/1000 centres, <= 11 flips; m = 4 x 32 bit): ONE engine
    (data + index built once), one short measurement per leg:
      exact    MIH_EXACT, 4096 near-duplicate queries per call            (search_worker.cc:159-218)
      exact16k the same in calls of 16384 queries (one launch: the launch's tail is paid once)
      uniform  MIH_EXACT, 64 uniform random queries per call: the radius loop would walk to shell ~8; answered through the
               cost-model switch by the verify kernel with the stop rule replayed
      approx   MIH_APPROX, 4096 near-duplicate queries per call           (search_worker.cc:93-157)
    Each leg: whole-call queries/s, kernel time, algorithmic bytes and `frac` of the HBM peak, a results check."""
    torch = env.torch
    from verticut_amd import engine as vc
    bits, m, k = 128, 4, 100
    rng = np.random.default_rng(args.seed + 3)
    out = {}
    t0 = time.perf_counter()
    e = vc.Engine(bits, capacity=n, n_tables=m, flags=vc.FLAG_LEAN_TIMING, timing_sample=MIH_TIMING_SAMPLE)
    try:
        e.add_synthetic(n, seed=args.seed, kind=vc.SYNTH_CLUSTERED, n_centres=max(n // 1000, 1), max_flips=11)
        e.build_index()
        env.sync()
        t_setup = time.perf_counter() - t0
        st = torch.cuda.current_stream().cuda_stream
        for leg in legs:
            name = "knn_%s_%s" % ({"exact": "mih", "exact16k": "mih", "uniform": "uniform", "approx": "approx"}[leg], tag) + ("_q16k" if leg == "exact16k" else "")
            try:
                Q = {"uniform": 64, "exact16k": 16384}.get(leg, 4096)
                mode = vc.MODE_MIH_APPROX if leg == "approx" else vc.MODE_MIH_EXACT
                if leg == "uniform":
                    host_q = [rng.integers(0, 256, size=(Q, bits // 8), dtype=np.uint8) for _ in range(2)]
                elif leg == "exact16k":     # one set, dealt two ways (a D2H code read per query: 16 384 of them take a second)
                    h0 = _near_queries(e, n, Q, bits, 4, rng)
                    host_q = [h0, np.ascontiguousarray(h0[::-1])]
                else:
                    host_q = [_near_queries(e, n, Q, bits, 4, rng) for _ in range(2)]
                dq = [torch.from_numpy(h).to(env.device) for h in host_q]
                d_out = torch.empty((Q, k), dtype=torch.int64, device=env.device)
                d_cnt = torch.empty((Q,), dtype=torch.int32, device=env.device)

                def one(i):
                    e.search_knn_dev(dq[i % 2].data_ptr(), Q, k, d_out.data_ptr(), d_cnt.data_ptr(), mode=mode, stream=st)

                nst = 4 if leg == "uniform" else steps
                for i in range(3):
                    one(i)
                env.sync()
                e.timing()
                _, elapsed = timed_steps(env, lambda c: [one(i) for i in range(c)], nst)
                tm = e.timing()
                got = d_out.cpu().numpy().view(np.uint64)
                cnt = d_cnt.cpu().numpy()
                qh = host_q[(nst - 1) % 2][:16]
                lin, _ = e.search_knn(qh, k, mode=vc.MODE_LINEAR)
                if leg == "approx":   # k genuine items per query, ascending, never nearer than the exact answer
                    ok = bool(np.all(cnt == k)) and bool(np.all(got[:, 1:] > got[:, :-1])) and \
                        bool(np.all((got[:16] >> np.uint64(32)) >= (lin >> np.uint64(32))))
                    for qi in range(2):
                        for j in (0, k // 2, k - 1):
                            code = e.get_code(int(got[qi, j] & np.uint64(0xFFFFFFFF)))
                            ok = ok and int(np.unpackbits(np.bitwise_xor(code, qh[qi])).sum()) == int(got[qi, j] >> np.uint64(32))
                else:                 # exact MIH distances == full scan
                    ok = bool(np.array_equal(got[:16] >> np.uint64(32), lin >> np.uint64(32)))
                e.timing()
                times = step_times_ms(env, one, nst)
                e.timing()
                r = {"workload": {"exact": "exact top-100 through MIH (MIH_EXACT), 128-bit, %.3g clustered codes, m=4, 4096 near-duplicate queries per call",
                                  "exact16k": "exact top-100 through MIH (MIH_EXACT), 128-bit, %.3g clustered codes, m=4, 16384 near-duplicate queries per call "
                                              "(one mih_query_kernel launch: a launch's tail -- it ends with its longest query -- is paid once per 16384 queries, not per 4096)",
                                  "uniform": "MIH_EXACT with 64 UNIFORM random queries per call over %.3g clustered codes: shells 0..3 in the query kernel, then the "
                                             "cost-model switch to the verify kernel, stop rule replayed",
                                  "approx": "approximate top-100 through MIH (MIH_APPROX, stop at 20k candidates), 128-bit, %.3g clustered codes, m=4, 4096 queries per call"}[leg] % n,
                     "value": Q * nst / elapsed, "unit": "queries/s", "ms_per_step": elapsed / nst * 1e3, "median_ms_per_step": _median(times),
                     "results_check": "ok" if ok else "FAILED"}
                r.update(_mih_fields(tm, bits, nst))
                if tm.scan_launches:      # the switch sent queries to the verify kernel: that kernel dominates the step
                    r["verify_kernel"] = _scan_fields(tm)
                    r["verify_kernel"]["kernel_ms_per_step"] = tm.scan_ms / nst
                    if leg == "uniform":
                        r["verify_kernel"]["note"] = ("the switch runs verify passes of up to 32 queries: per query the VALU-bound pass of 32 is cheaper than the "
                                                      "HBM-bound pass of 8 (extras.qt32 vs the headline), so `frac` -- the pass's share of the HBM peak -- is ~0.25 by design")
                        r["mih_kernel_frac"] = r["frac"]
                        r["frac"] = r["verify_kernel"]["frac"]
                        r["kernel"] = "vc_scan_kernel (dominant: %.3f of %.3f ms per step) behind mih_query_kernel" % (tm.scan_ms / nst, elapsed / nst * 1e3)
                if leg in ("exact", "exact16k"):
                    # what a SearchWorker::find caller sees: the same calls through vc_search_knn with HOST pointers --
                    # H2D of the queries, D2H of Q x k x 8 bytes of rows into pageable memory, inside the timed region
                    hp = {"rows_bytes_per_call": Q * k * 8,
                          "what": "vc_search_knn: H2D of the queries + D2H of the rows inside the timed region, synchronous calls; result buffers "
                                  "allocated once -- ordinary (pageable) numpy arrays, and page-locked ones (one DMA, no staging)"}
                    for kind in ("pageable", "pinned"):
                        if kind == "pinned":
                            h_out = torch.empty((Q, k), dtype=torch.int64).pin_memory().numpy().view(np.uint64)
                            h_cnt = torch.empty((Q,), dtype=torch.int32).pin_memory().numpy().view(np.uint32)
                            h_q = [torch.from_numpy(h).pin_memory().numpy() for h in host_q]
                        else:
                            h_out, h_cnt, h_q = np.zeros((Q, k), dtype=np.uint64), np.zeros(Q, dtype=np.uint32), host_q
                        e.search_knn(h_q[0], k, mode=mode, out=h_out, counts=h_cnt)
                        th = time.perf_counter()
                        for i in range(4):
                            e.search_knn(h_q[i % 2], k, mode=mode, out=h_out, counts=h_cnt)
                        dth = time.perf_counter() - th
                        ok = ok and bool(np.array_equal(h_out[:16] >> np.uint64(32), e.search_knn(h_q[3 % 2][:16], k, mode=vc.MODE_LINEAR)[0] >> np.uint64(32)))
                        hp[kind] = {"value": Q * 4 / dth, "unit": "queries/s", "ms_per_call": dth / 4 * 1e3}
                    e.timing()
                    r["host_pointers"] = hp
                    r["results_check"] = "ok" if ok else "FAILED"
                if leg == "exact":
                    r["setup_s"] = t_setup        # data generation + index build (+ {id, code} records)
                out[name] = r
            except Exception as ex:
                out[name] = {"error": "%s: %s" % (type(ex).__name__, ex)}
    finally:
        e.close()
    return out


def run_extras(args, env):
    out = {}
    for name, fn in (("qt32", lambda: _extra_qt32(args, env)), ("c2_m2", lambda: _extra_c2(args, env))):
        try:
            out[name] = fn()
        except Exception as ex:   # an extra never takes the headline down with it; the failure is reported in its place
            out[name] = {"error": "%s: %s" % (type(ex).__name__, ex)}
    for n, tag, legs in ((100_000_000, "1e8", ("exact", "exact16k", "approx")), (1_000_000_000, "1e9", ("exact", "exact16k", "uniform"))):
        try:
            out.update(_extras_knn_mih(args, env, n, tag, legs))
        except Exception as ex:
            out["knn_mih_%s" % tag] = {"error": "%s: %s" % (type(ex).__name__, ex)}
    return out


def _scan_or_mih_roofline(tm, bits):
    """knn_mih: the MIH query kernel's record, plus the verify kernel's when the cost-model switch sent queries its way"""
    r = _mih_roofline(tm, bits)
    if tm.scan_launches:
        avg = tm.scan_ms / tm.scan_launches
        r["verify_kernel"] = {"kernel": "vc_scan_kernel", "launches": tm.scan_launches, "avg_launch_ms": avg,
                              "algorithmic_bytes_per_launch": tm.scan_bytes / tm.scan_launches,
                              "achieved": tm.scan_bytes / tm.scan_launches / (avg * 1e-3) / 1e9 if avg > 0 else 0.0, "unit": "GB/s",
                              "frac": tm.scan_bytes / tm.scan_launches / (avg * 1e-3) / 1e9 / HBM_PEAK_GBPS if avg > 0 else 0.0}
    return r


def run_c2(args, env, emit):
    """BASELINE configs[1]: 64-bit codes, 1e8 DB, all neighbours within distance 8 through MIH (search_worker.cc:222-264)."""
    torch = env.torch
    from verticut_amd import engine as vc
    n, bits, Q, radius = int(args.n), args.bits, args.queries, 8
    rng = np.random.default_rng(args.seed + 2)
    lines = {}
    tables = [int(t) for t in args.tables.split(",")]
    for m in tables:
        e = vc.Engine(bits, capacity=n, n_tables=m, flags=vc.FLAG_LEAN_TIMING, timing_sample=MIH_TIMING_SAMPLE)
        e.add_synthetic(n, seed=args.seed)
        t0 = time.perf_counter()
        e.build_index()
        t_build = time.perf_counter() - t0
        host_q = [_near_queries(e, n, Q, bits, radius, rng) for _ in range(2)]
        dev_q = [torch.from_numpy(h).to(env.device) for h in host_q]
        out_cap = Q * 64
        d_out = torch.empty((out_cap,), dtype=torch.int64, device=env.device)
        d_off = torch.empty((Q + 1,), dtype=torch.int64, device=env.device)
        s = torch.cuda.current_stream().cuda_stream

        def run_steps(count):
            for i in range(count):
                rc = e.search_radius_dev(dev_q[i % 2].data_ptr(), Q, radius, d_out.data_ptr(), out_cap, d_off.data_ptr(),
                                         mode=vc.MODE_MIH_EXACT, stream=s)
                if rc != vc.VC_OK:      # VC_ERR_CAPACITY: a truncated result must never be timed as if it were complete
                    raise SystemExit("c2: search_radius_dev returned %d (results do not fit out_cap = %d)" % (rc, out_cap))

        run_steps(max(args.warmup, 2))
        env.sync()
        e.timing()
        _, elapsed = timed_steps(env, run_steps, args.steps)
        tm = e.timing()
        ok = True
        if not args.no_check:   # MIH == full scan on the last batch (the oracle-level parity lives in tests/)
            off = d_off.cpu().numpy().view(np.uint64)
            res = d_out.cpu().numpy().view(np.uint64)
            qh = host_q[(args.steps - 1) % 2]
            lin = e.search_radius(qh[:32], radius, mode=vc.MODE_LINEAR)
            for i in range(32):
                ok = ok and np.array_equal(res[int(off[i]):int(off[i + 1])], lin[i])
            if not ok:
                sys.stderr.write("[bench] c2 m=%d: MIH result differs from the linear scan\n" % m)
        lines[m] = {
            "value": Q * args.steps / elapsed, "ms_per_step": elapsed / args.steps * 1e3,
            "roofline": _mih_roofline(tm, bits, "mih_bucket_stream_kernel" if bits // m <= 16 else "mih_query_kernel"),
            "index_build_s": t_build, "mean_neighbours": float(d_off[Q].item()) / Q, "results_check": "ok" if ok else "FAILED",
        }
        e.close()
    ok = all(v["results_check"] == "ok" for v in lines.values())
    m0 = tables[0]
    line = {
        "metric": "queries/sec (all neighbours within Hamming distance 8, MIH) on 64-bit codes, 100M DB; bit-exact vs linear scan",
        "value": lines[m0]["value"], "unit": "queries/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": lines[m0]["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u64", "data": "synthetic",
        "config": {
            "workload": "BASELINE configs[1]: %d-bit codes, %.3g-code DB, MIH r=8 neighbour search, m=%d x %d-bit substrings "
                        "(value)%s" % (bits, n, m0, bits // m0, "; m=4 x 16-bit as variant" if len(tables) > 1 else ""),
            "n_codes": n, "bits": bits, "radius": radius, "queries_per_step": Q, "seed": args.seed,
            "query_kind": "DB item with 0-8 random bit flips (query-by-image-id use, image_search_client.h:23-25)",
            "api": "vc_search_radius_dev: queries, results and offsets resident in HBM",
            "index_build_s": lines[m0]["index_build_s"], "mean_neighbours_per_query": lines[m0]["mean_neighbours"],
            "variants": {"m%d_s%d" % (m, bits // m): {"value": lines[m]["value"], "ms_per_step": lines[m]["ms_per_step"],
                                                      "roofline": lines[m]["roofline"], "results_check": lines[m]["results_check"]}
                         for m in tables[1:]},
        },
        "roofline": lines[m0]["roofline"],
        "results_check": "ok" if ok else "FAILED",
    }
    if not args.no_traffic and line["roofline"].get("achieved") is not None:
        # memory-side requests of the query kernel (64 bytes each); the x2 of a wide stream does not apply to 16-byte gathers
        t, how = measure_traffic(args, "mih_query_kernel", fetch_mult=1.0, extra=("--tables", str(m0)))
        line["roofline"]["traffic"], line["roofline"]["traffic_how"] = t, how + " (64-byte requests of 16-byte granule loads and gathers; uncorrected)"
        if lines[m0]["roofline"].get("kernel") == "mih_query_kernel":
            _add_sector_roofline(line["roofline"])
    if args.cpu_seconds > 0:
        line["cpu_baseline"] = cpu_baseline_mih(args, m0, "radius", radius=radius)
    emit(line)
    return ok


def run_knn_mih(args, env, emit):
    """exact top-k through MIH (SearchWorker::find, search_worker.cc:65-89,159-218) on clustered codes"""
    torch = env.torch
    from verticut_amd import engine as vc
    n, bits, Q, k, m = int(args.n), args.bits, args.queries, args.k, 4
    rng = np.random.default_rng(args.seed + 3)
    e = vc.Engine(bits, capacity=n, n_tables=m, flags=vc.FLAG_LEAN_TIMING, timing_sample=MIH_TIMING_SAMPLE)
    e.add_synthetic(n, seed=args.seed, kind=vc.SYNTH_CLUSTERED, n_centres=max(n // 1000, 1), max_flips=11)
    e.build_index()
    if args.uniform_queries:
        host_q = [rng.integers(0, 256, size=(Q, bits // 8), dtype=np.uint8) for _ in range(2)]
    else:
        host_q = [_near_queries(e, n, Q, bits, 4, rng) for _ in range(2)]
    dev_q = [torch.from_numpy(h).to(env.device) for h in host_q]
    d_out = torch.empty((Q, k), dtype=torch.int64, device=env.device)
    d_cnt = torch.empty((Q,), dtype=torch.int32, device=env.device)
    s = torch.cuda.current_stream().cuda_stream

    mode = vc.MODE_MIH_APPROX if args.approximate else vc.MODE_MIH_EXACT

    def one(i):
        e.search_knn_dev(dev_q[i % 2].data_ptr(), Q, k, d_out.data_ptr(), d_cnt.data_ptr(), mode=mode, stream=s)

    def run_steps(count):
        for i in range(count):
            one(i)

    run_steps(max(args.warmup, 2))
    env.sync()
    e.timing()
    _, elapsed = timed_steps(env, run_steps, args.steps)
    tm = e.timing()
    ok = True
    if not args.no_check:   # exact MIH == full scan on the distances (ids may differ among ties at the k-th distance)
        got = d_out.cpu().numpy().view(np.uint64)
        lin, _ = e.search_knn(host_q[(args.steps - 1) % 2][:16], k, mode=vc.MODE_LINEAR)
        if args.approximate:   # k genuine items per query, ascending, never nearer than the exact answer
            ok = bool(np.all(d_cnt.cpu().numpy() == k)) and bool(np.all(got[:, 1:] > got[:, :-1])) and \
                bool(np.all((got[:16] >> np.uint64(32)) >= (lin >> np.uint64(32))))
        else:
            ok = bool(np.array_equal(got[:16] >> np.uint64(32), lin >> np.uint64(32)))
    e.timing()
    step_ms = step_times_ms(env, one, args.steps)
    e.close()
    line = {
        "metric": "queries/sec (%s k-NN top-%d through MIH) on %d-bit clustered codes, %.3g DB; %s" %
                  ("approximate" if args.approximate else "exact", k, bits, n,
                   "search_worker.cc:93-157 restated (parity in tests/)" if args.approximate else "distances bit-exact vs linear scan"),
        "value": Q * args.steps / elapsed, "unit": "queries/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "median_ms_per_step": _median(step_ms), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u64", "data": "synthetic",
        "config": {
            "workload": "SearchWorker::find %s MIH: %d-bit codes, %.3g clustered codes (n/1000 centres, <= 11 flips), m=4 x 32-bit, top-%d"
                        % ("approximate" if args.approximate else "exact", bits, n, k),
            "n_codes": n, "bits": bits, "k": k, "queries_per_step": Q, "seed": args.seed,
            "query_kind": "uniform random (radius loop would need shells up to r ~ 8: answered by the verify kernel through the cost-model "
                          "switch, stop rule replayed)" if args.uniform_queries else "DB item with 0-4 random bit flips",
            "api": "vc_search_knn_dev: queries and results resident in HBM",
        },
        "roofline": _scan_or_mih_roofline(tm, bits),
        "results_check": "ok" if ok else "FAILED",
    }
    if not args.no_traffic:
        t, how = measure_traffic(args, "mih_query_kernel", fetch_mult=1.0,
                                 extra=(["--approximate"] if args.approximate else []) + (["--uniform-queries"] if args.uniform_queries else []))
        line["roofline"]["traffic"], line["roofline"]["traffic_how"] = t, how + " (64-byte requests of 16-byte granule loads and gathers; uncorrected)"
        if line["roofline"].get("kernel") == "mih_query_kernel":
            _add_sector_roofline(line["roofline"])
    if args.cpu_seconds > 0:
        line["cpu_baseline"] = cpu_baseline_mih(args, m, "knn", clustered=True, approximate=args.approximate)
    emit(line)
    return ok


def run_c1(args, env, emit):
    """BASELINE configs[0] / BASELINE.md row C1: linear_search.cc:39-64 over 64-bit codes, N = 2^20, the reference's 200-query
    cap per run (distributed_image_search.cc:83-84), k = 100 -- the plumbing shape.  The GPU value stands next to BASELINE.md's
    CPU-linear-1T and CPU-linear-allcores rows (cpu_baseline / cpu_baseline.allcores), which run on the WHOLE database here."""
    torch = env.torch
    from verticut_amd import engine as vc
    n, bits, Q, k = int(args.n), args.bits, args.queries, args.k
    rng = np.random.default_rng(args.seed + 1)
    e = vc.Engine(bits, capacity=n, flags=vc.FLAG_LEAN_TIMING)      # query tile left to the engine: 8 MB of codes -> the 200 queries of a call in ONE pass
    e.add_synthetic(n, seed=args.seed)
    host_q = [rng.integers(0, 256, size=(Q, bits // 8), dtype=np.uint8) for _ in range(2)]
    dev_q = [torch.from_numpy(h).to(env.device) for h in host_q]
    d_out = torch.empty((Q, k), dtype=torch.int64, device=env.device)
    d_cnt = torch.empty((Q,), dtype=torch.int32, device=env.device)
    s = torch.cuda.current_stream().cuda_stream

    def one(i):
        e.search_knn_dev(dev_q[i % 2].data_ptr(), Q, k, d_out.data_ptr(), d_cnt.data_ptr(), stream=s)

    def run_steps(count):
        for i in range(count):
            one(i)

    run_steps(max(args.warmup, 2))
    env.sync()
    e.timing()
    _, elapsed = timed_steps(env, run_steps, args.steps)
    tm = e.timing()
    ok = True
    if not args.no_check:
        res = d_out.cpu().numpy().view(np.uint64)
        qh = host_q[(args.steps - 1) % 2]
        ok = bool(np.all(d_cnt.cpu().numpy() == min(k, n))) and bool(np.all(res[:, 1:] > res[:, :-1]))
        for qi in (0, Q // 2, Q - 1):
            for j in (0, k - 1):
                code = e.get_code(int(res[qi, j] & np.uint64(0xFFFFFFFF)))
                ok = ok and int(np.unpackbits(np.bitwise_xor(code, qh[qi])).sum()) == int(res[qi, j] >> np.uint64(32))
        ok = ok and e.device_status() == 0
    e.timing()
    step_ms = step_times_ms(env, one, args.steps)
    # the host-pointer form of the same step (what the reference's driver loop does per query, distributed_image_search.cc:62-85)
    t0 = time.perf_counter()
    host_steps = max(2, min(args.steps, 10))
    for i in range(host_steps):
        e.search_knn(host_q[i % 2], k)
    host_elapsed = time.perf_counter() - t0
    e.close()
    sf = _scan_fields(tm)
    line = {
        "metric": "queries/sec (k-NN top-%d, linear_search) on %d-bit codes, %d-code DB; bit-exact vs linear_search" % (k, bits, n),
        "value": Q * args.steps / elapsed, "unit": "queries/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "median_ms_per_step": _median(step_ms), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {
            "workload": "BASELINE configs[0]: linear_search.cc brute-force k-NN, %d-bit codes, %d synthetic images (2^20), %d queries per "
                        "run (the reference's cap), top-%d; the reference runs this shape on the CPU only -- its rows are cpu_baseline" % (bits, n, Q, k),
            "n_codes": n, "bits": bits, "k": k, "queries_per_step": Q, "query_tile": "left to the engine (vc_config.query_tile = 0): 8 MB of codes -> up to 512 queries per pass, i.e. the 200 queries of a call in one pass", "query_kind": "uniform random", "seed": args.seed,
            "api": "vc_search_knn_dev: queries and results resident in HBM",
            "host_pointer_api": {"value": Q * host_steps / host_elapsed, "unit": "queries/s", "ms_per_step": host_elapsed / host_steps * 1e3,
                                 "what": "the same step through vc_search_knn (H2D of the queries, D2H of the rows, synchronous)"},
        },
        "roofline": {
            "bound": "hbm", "achieved": sf["achieved_GBps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": sf["frac"], "traffic": None,
            "kernel": "vc_scan_kernel", "launches": tm.scan_launches, "avg_launch_ms": sf["kernel_avg_ms"],
            "algorithmic_bytes_per_launch": sf["algorithmic_bytes_per_launch"],
            "note": "an 8 MB database lives in L2 / the Infinity Cache and a pass of 32 queries over it is launch- and VALU-bound: "
                    "the HBM fraction of this plumbing shape says nothing about the kernel (the roofline run is configs[2])",
        },
        "results_check": "ok" if ok else "FAILED",
    }
    if args.cpu_seconds > 0:
        line["cpu_baseline"] = cpu_baseline_linear(args, n)
    emit(line)
    return ok


def run_sharded1dev(args, env, emit):
    """BASELINE configs[3]'s shard arithmetic behind the C ABI in ONE process on ONE device: the 1e9-code database as
    `--shards` id-range shards through vc_sharded_search_knn_dev (queries, rows and counts resident in HBM, one stream).
    Reported: ms per batch through the sharded layer against (a) the sum of the shards' verify-kernel times and (b) the
    same shard engines driven back to back WITHOUT the layer (vc_search_knn_dev per shard, no exchange, no merge) -- the
    difference to (b) is what the layer itself costs per batch (its host work, the merge launch, the slot bookkeeping)."""
    torch = env.torch
    from verticut_amd import engine as vc
    n, bits, Q, k, G = int(args.n), args.bits, args.queries, args.k, args.shards
    rng = np.random.default_rng(args.seed + 1)
    tsample = int(os.environ.get("VC_BENCH_TIMING_SAMPLE", "8"))     # an event pair per verify launch costs ~8 us: time every 8th
    s = vc.ShardedEngine(bits, capacity=n, n_shards=G, devices=[0], query_tile=Q, timing_sample=tsample)
    s.add_synthetic(n, seed=args.seed)
    shards = [s.shard(g) for g in range(G)]
    host_q = [rng.integers(0, 256, size=(Q, bits // 8), dtype=np.uint8) for _ in range(4)]
    dev_q = [torch.from_numpy(h).to(env.device) for h in host_q]
    d_out = torch.empty((Q, k), dtype=torch.int64, device=env.device)
    d_cnt = torch.empty((Q,), dtype=torch.int32, device=env.device)
    raw = torch.empty((G, Q, k), dtype=torch.int64, device=env.device)
    rcnt = torch.empty((G, Q), dtype=torch.int32, device=env.device)
    st = torch.cuda.current_stream().cuda_stream

    def one(i):
        s.search_knn_dev(dev_q[i % 4].data_ptr(), Q, k, d_out.data_ptr(), d_cnt.data_ptr(), stream=st)

    def bare(i):          # the floor: the same engines, the same stream, no sharded layer
        for g, e in enumerate(shards):
            e.search_knn_dev(dev_q[i % 4].data_ptr(), Q, k, raw[g].data_ptr(), rcnt[g].data_ptr(), stream=st)

    for i in range(max(args.warmup, 8)):
        one(i)
    env.sync()
    for e in shards:
        e.timing()
    _, elapsed = timed_steps(env, lambda c: [one(i) for i in range(c)], args.steps)
    tms = [e.timing() for e in shards]
    res = d_out.cpu().numpy().view(np.uint64).copy()
    cnt = d_cnt.cpu().numpy().copy()
    step_ms = step_times_ms(env, one, args.steps)
    # host time spent ENQUEUEING a batch (the call returns before the device has finished)
    env.sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        one(i)
    host_issue_ms = (time.perf_counter() - t0) / args.steps * 1e3
    env.sync()
    for i in range(4):
        bare(i)
    env.sync()
    _, bare_elapsed = timed_steps(env, lambda c: [bare(i) for i in range(c)], args.steps)
    # the two forms alternate twice more: the layer's cost is a difference of two ~3 ms figures, a drifting clock shows in it
    pairs = []
    for _ in range(2):
        _, a = timed_steps(env, lambda c: [one(i) for i in range(c)], args.steps)
        _, b = timed_steps(env, lambda c: [bare(i) for i in range(c)], args.steps)
        pairs.append(((a - b) / args.steps * 1e6, a / args.steps * 1e3, b / args.steps * 1e3))
    ok = True
    if not args.no_check:    # the sharded rows == merge of the bare shard rows of the same batch, distances recomputed from the stored codes
        qi = (args.steps - 1) % 4
        bare(qi)
        merged = torch.empty((Q, k), dtype=torch.int64, device=env.device)
        vc.merge_topk_dev(raw.data_ptr(), G, Q, k, merged.data_ptr(), None, stream=st)
        env.sync()
        ok = bool(np.array_equal(res, merged.cpu().numpy().view(np.uint64))) and bool(np.all(cnt == k)) and bool(np.all(res[:, 1:] > res[:, :-1]))
        for j in (0, k - 1):
            code = s.get_code(int(res[0, j] & np.uint64(0xFFFFFFFF)))
            ok = ok and int(np.unpackbits(np.bitwise_xor(code, host_q[qi][0])).sum()) == int(res[0, j] >> np.uint64(32))
    for e in shards:
        e.close()
    s.close()
    scan_avgs = [t.scan_ms / max(t.scan_launches, 1) for t in tms]
    sum_scan = float(sum(scan_avgs))
    ms = elapsed / args.steps * 1e3
    bare_ms = bare_elapsed / args.steps * 1e3
    ach = n * (bits // 8) / (sum_scan * 1e-3) / 1e9 if sum_scan > 0 else 0.0
    line = {
        "metric": METRIC, "value": Q * args.steps / elapsed, "unit": "queries/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms, "median_ms_per_step": _median(step_ms), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "u64", "data": "synthetic",
        "config": {
            "workload": "BASELINE configs[3]'s partition on ONE device: %d-bit codes, %.3g-code DB as %d id-range shards behind vc_sharded_search_knn_dev "
                        "(one process, one stream), top-%d, per-shard top-k merged on the device" % (bits, n, G, k),
            "n_codes": n, "bits": bits, "k": k, "queries_per_step": Q, "shards": G, "query_kind": "uniform random", "seed": args.seed,
            "api": "vc_sharded_search_knn_dev: queries, rows and counts resident in HBM",
        },
        "sharded_layer": {
            "ms_per_batch": ms, "sum_of_shard_scan_kernel_ms": sum_scan, "shard_scan_kernel_ms": scan_avgs,
            "same_engines_without_the_layer_ms": bare_ms, "layer_overhead_us": (ms - bare_ms) * 1e3,
            "layer_overhead_us_repeats": [round(p[0], 1) for p in pairs],   # two more alternating (layer, bare) pairs of `steps` batches
            "layer_overhead_us_median": float(np.median([(ms - bare_ms) * 1e3] + [p[0] for p in pairs])),
            "around_the_verify_kernels_us": (ms - sum_scan) * 1e3, "host_enqueue_ms_per_batch": host_issue_ms,
            "what": "layer_overhead_us = ms_per_batch - the same %d shard engines driven back to back on the same stream without exchange / merge; "
                    "around_the_verify_kernels_us also holds every shard's bootstrap / select / recover launches" % G,
        },
        "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS, "traffic": None,
                     "kernel": "vc_scan_kernel", "launches": sum(t.scan_launches for t in tms), "avg_launch_ms": sum_scan / max(G, 1),
                     "algorithmic_bytes_per_launch": n * (bits // 8) / max(G, 1),
                     "note": "all %d shards' verify launches of a batch together read the database once: achieved = N x B/8 / their summed time" % G},
        "results_check": "ok" if ok else "FAILED",
    }
    emit(line)
    return ok


def run_c5shard(args, env, emit):
    """one GPU's share of BASELINE configs[4]: 256-bit codes, 5e8 of the 4e9 codes, 4096 queries per pass (LDS query tile)"""
    torch = env.torch
    from verticut_amd import engine as vc
    n, bits, Q, k = int(args.n), args.bits, args.queries, args.k
    rng = np.random.default_rng(args.seed + 5)
    e = vc.Engine(bits, capacity=n, query_tile=Q, flags=vc.FLAG_LEAN_TIMING)
    e.add_synthetic(n, seed=args.seed)
    host_q = [rng.integers(0, 256, size=(Q, bits // 8), dtype=np.uint8) for _ in range(2)]
    dev_q = [torch.from_numpy(h).to(env.device) for h in host_q]
    d_out = torch.empty((Q, k), dtype=torch.int64, device=env.device)
    d_cnt = torch.empty((Q,), dtype=torch.int32, device=env.device)
    s = torch.cuda.current_stream().cuda_stream

    def run_steps(count):
        for i in range(count):
            e.search_knn_dev(dev_q[i % 2].data_ptr(), Q, k, d_out.data_ptr(), d_cnt.data_ptr(), stream=s)

    run_steps(max(1, min(args.warmup, 2)))
    env.sync()
    e.timing()
    _, elapsed = timed_steps(env, run_steps, args.steps)
    tm = e.timing()
    ok = True
    if not args.no_check:
        res = d_out.cpu().numpy().view(np.uint64)
        qh = host_q[(args.steps - 1) % 2]
        ok = bool(np.all(d_cnt.cpu().numpy() == k)) and bool(np.all(res[:, 1:] > res[:, :-1]))
        for qi in (0, Q // 2, Q - 1):
            for j in (0, k - 1):
                code = e.get_code(int(res[qi, j] & np.uint64(0xFFFFFFFF)))
                ok = ok and int(np.unpackbits(np.bitwise_xor(code, qh[qi])).sum()) == int(res[qi, j] >> np.uint64(32))
        ok = ok and e.device_status() == 0
    e.close()
    launches = max(tm.scan_launches, 1)
    avg_ms = tm.scan_ms / launches
    ops = Q * n * (bits // 32) * 2            # Q x N x (B/32 xor + B/32 popcount)
    achieved = ops / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    line = {
        "metric": "queries/sec (k-NN top-%d) on %d-bit codes, %.3g-code shard, %d queries per pass; bit-exact vs linear_search" % (k, bits, n, Q),
        "value": Q * args.steps / elapsed, "unit": "queries/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u64", "data": "synthetic",
        "config": {
            "workload": "one GPU's share of BASELINE configs[4]: %d-bit codes, %.3g of the 4e9 codes, %d queries batched in one "
                        "LDS query tile, top-%d, linear verify kernel" % (bits, n, Q, k),
            "n_codes": n, "bits": bits, "k": k, "queries_per_step": Q, "query_tile": Q, "query_kind": "uniform random", "seed": args.seed,
        },
        "roofline": {
            "bound": "valu", "achieved": achieved, "peak": VALU_PEAK_GOPS, "unit": "Gop/s", "frac": achieved / VALU_PEAK_GOPS,
            "traffic": None, "kernel": "vc_scan_kernel", "launches": tm.scan_launches, "avg_launch_ms": avg_ms,
            "algorithmic_ops_per_launch": ops,
            "hbm": {"achieved": tm.scan_bytes / launches / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0, "peak": HBM_PEAK_GBPS, "unit": "GB/s"},
            "note": "4096 queries per database pass put the verify kernel two orders of magnitude on the VALU side of the "
                    "roofline (SURVEY.md 7.2): the bound is the xor + popcount issue rate, not HBM; no MFMA by design",
        },
        "results_check": "ok" if ok else "FAILED",
    }
    if args.cpu_seconds > 0:
        line["cpu_baseline"] = cpu_baseline_linear(args, n)
    emit(line)
    return ok


def main(argv=None, backend_factory=None, device_kind="cuda", dist_backend=None, out=None):
    """backend_factory / device_kind / dist_backend / out exist for the CPU rehearsal of the multi-rank flow
    (tests/test_bench_cpu.py: world size 2 over gloo with a test double for the GPU backend)."""
    args = parse(argv)
    if out is None:
        # The contract is ONE JSON line on stdout.  Native libraries print there too (RCCL writes a five-line version
        # banner when its communicator comes up), so fd 1 is pointed at stderr for the whole run and the JSON line goes
        # to a private duplicate of the original stdout.
        sys.stdout.flush()
        out = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)

    def emit(line):
        out.write(json.dumps(line) + "\n")
        out.flush()

    env = Env(args, device_kind=device_kind, dist_backend=dist_backend)
    if args.workload != "c3" and env.world > 1:
        raise SystemExit("--workload %s is a single-GPU line" % args.workload)
    try:
        if args.workload == "c3":
            ok = run_headline(args, env, emit, backend_factory=backend_factory)
        elif args.workload == "c1":
            ok = run_c1(args, env, emit)
        elif args.workload == "sharded1dev":
            ok = run_sharded1dev(args, env, emit)
        elif args.workload == "c2":
            ok = run_c2(args, env, emit)
        elif args.workload == "knn_mih":
            ok = run_knn_mih(args, env, emit)
        else:
            ok = run_c5shard(args, env, emit)
    finally:
        env.close()
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
