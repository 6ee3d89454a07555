#!/usr/bin/env python3
"""bench.py -- BASELINE.json headline: queries/sec, k-NN top-100, 128-bit codes, 1B-code database
(configs[2]; at --gpus N the same 1B database is sharded N ways = configs[3], strong scaling).

One step = one batch of Q queries verified against the whole database by the HBM-bound verify kernel
(vc_scan_kernel) + top-k select (+ all-gather/merge of per-shard top-k when N > 1).  Database, queries and
result buffers are resident in HBM before the timed region starts.

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` (achieved HBM GB/s of the
verify kernel from HIP events recorded on its stream, against the 8 TB/s peak) and `cpu_baseline` (the CPU
restatement of linear_search.cc timed on this host on a bounded sample).

    python bench.py                               # 1 GPU, 1e9 codes
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md; ~6.3 TB/s is what a pure copy reaches)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--db-size", dest="n", type=float, default=1e9, help="database size (total over all GPUs)")
    ap.add_argument("--bits", type=int, default=128)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--queries", type=int, default=8, help="queries per step (= one query tile = one DB pass)")
    ap.add_argument("--seed", type=int, default=34, help="linear_search.cc:105 srand(34)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--no-check", action="store_true")
    return ap.parse_args()


def cpu_baseline(args, n_total):
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
    res = {
        "value": qps_sample * sample_n / n_total,
        "unit": "queries/s",
        "cores": 1,
        "kind": "port",
        "sample": "%d queries x first %d codes of the same synthetic DB in %.1f s, 1 thread; scaled by N_sample/N "
                  "(scan cost is linear in N); structure of linear_search.cc:39-64 over in-memory codes (no KV get)"
                  % (done, sample_n, dt),
        "items_per_s": qps_sample * sample_n,
    }
    # all host cores (BASELINE.md row CPU-linear-allcores): same scan split over threads + merge
    cores = min(os.cpu_count() or 1, 16)   # this GPU's share of the host (one of eight GPUs on the node)
    t0 = time.perf_counter()
    done = 0
    while done < len(q) and time.perf_counter() - t0 < max(2.0, args.cpu_seconds / 3):
        vo.linear_knn(codes, q[done], args.k, threads=cores)
        done += 1
    dt = time.perf_counter() - t0
    res["allcores"] = {"value": done / dt * sample_n / n_total, "cores": cores}
    return res


def main():
    args = parse()
    # The contract is ONE JSON line on stdout.  Native libraries print there too (RCCL writes a five-line version
    # banner when its communicator comes up), so fd 1 is pointed at stderr for the whole run and the JSON line goes to
    # a private duplicate of the original stdout.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    from verticut_amd import engine as vc
    from verticut_amd.sharded import ShardedSearch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with %d ranks" % (args.gpus, args.gpus))
        args.gpus = world
    # rehearsal knobs (dev only): run the multi-rank flow on ONE GPU, where RCCL cannot be used (one device per rank)
    backend = os.environ.get("VC_BENCH_BACKEND", "nccl")
    if os.environ.get("VC_BENCH_ONE_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    n_total = int(args.n)
    Q, k = args.queries, args.k
    force_exchange = world == 1 and os.environ.get("VC_BENCH_FORCE_EXCHANGE") == "1"   # dev: RCCL exchange with one rank
    if force_exchange:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        dist.init_process_group(backend, rank=0, world_size=1, **({"device_id": device} if backend == "nccl" else {}))
    engine_kw = {}
    if os.environ.get("VC_BENCH_SCAN_BLOCKS"):      # dev: cap the persistent verify grid (leave block slots to other kernels)
        engine_kw["scan_blocks"] = int(os.environ["VC_BENCH_SCAN_BLOCKS"])
    # N > 1: the per-shard top-k of 8 consecutive batches share one all-gather + merge (ShardedSearch(bucket=8)): the
    # collective is latency-bound at 6.4 KB per rank, every step still ends inside the timed region (flush()).
    bucket = int(os.environ.get("VC_BENCH_BUCKET", "8")) if (world > 1 or force_exchange) else 1
    ss = ShardedSearch(args.bits, n_total, rank=rank, world=world, device=local_rank, query_tile=Q,
                       force_exchange=force_exchange, bucket=bucket, **engine_kw)
    ss.add_synthetic(args.seed)

    # query batches resident in HBM: uniform random codes = worst case (no early threshold help)
    rng = np.random.default_rng(args.seed + 1)
    nb = 4
    host_q = [rng.integers(0, 256, size=(Q, args.bits // 8), dtype=np.uint8) for _ in range(nb)]
    dev_q = [torch.from_numpy(h).to(device) for h in host_q]
    torch.cuda.synchronize()

    def run_steps(count):
        res = None
        for i in range(count):
            res = ss.search(dev_q[i % nb], k)
        ss.flush()                  # N > 1, pipelined exchange: the last batches' side-stream work joins this stream
        return res

    # Priming, part of set-up like the data generation above: the first launches after the 16 GB fill run 10-15 %
    # slow (clocks, page tables; kernel trace in profiles/), and a driver-chosen --warmup may be shorter than that.
    run_steps(8)
    torch.cuda.synchronize()
    run_steps(args.warmup)
    torch.cuda.synchronize()
    exchange = "none"
    if world > 1 or force_exchange:
        # The per-batch exchange (all-gather + merge) runs inline on the step's stream: the plain, widely used pattern.
        # VC_BENCH_PIPELINED=1 moves it to a side stream under the next batch's scan (ShardedSearch(pipelined=True));
        # that path is covered by tests but has never run over RCCL on a multi-GPU node, so it is opt-in.
        ss.pipelined = os.environ.get("VC_BENCH_PIPELINED") == "1"
        exchange = "side-stream" if ss.pipelined else ("inline, %d batches per all-gather" % bucket if bucket > 1 else "inline")
        if ss.pipelined:
            run_steps(2)
            torch.cuda.synchronize()
    ss.backend.timing()  # drop warm-up event records

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out, cnt = run_steps(args.steps)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    if world > 1:
        dist.barrier()
    elapsed = t1 - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    tm = ss.backend.timing()  # HIP events on the launch stream, exactly the timed steps

    ok = True
    if not args.no_check:
        # size-independent properties of the last batch (bit-exact parity proper lives in tests/): ascending
        # packed values, k results, every reported distance recomputed from the stored code.
        res = out.cpu().numpy().view(np.uint64)
        qh = host_q[(args.steps - 1) % nb]
        why = []
        if not np.all(cnt.cpu().numpy() == min(k, n_total)):
            why.append("counts %s" % cnt.cpu().numpy().tolist())
        if not np.all(res[:, 1:] > res[:, :-1]):
            why.append("rows not strictly ascending")
        for qi in range(min(Q, 2)):
            for j in (0, k // 2, k - 1):
                gid = int(res[qi, j] & np.uint64(0xFFFFFFFF))
                if ss.lo <= gid < ss.hi:
                    code = ss.backend.engine.get_code(gid)
                    d = int(np.unpackbits(np.bitwise_xor(code, qh[qi])).sum())
                    if d != int(res[qi, j] >> np.uint64(32)):
                        why.append("query %d result %d: id %d reported %d, stored code says %d"
                                   % (qi, j, gid, int(res[qi, j] >> np.uint64(32)), d))
        ok = not why
        if why:
            sys.stderr.write("[bench rank %d] results check failed: %s\n" % (rank, "; ".join(why[:4])))
        if world > 1:
            okt = torch.tensor([1 if ok else 0], device=device)
            dist.all_reduce(okt, op=dist.ReduceOp.MIN)
            ok = bool(okt.item())

    if rank == 0:
        scan_avg_ms = tm.scan_ms / max(tm.scan_launches, 1)
        bytes_per_launch = tm.scan_bytes / max(tm.scan_launches, 1)
        achieved = bytes_per_launch / (scan_avg_ms * 1e-3) / 1e9 if scan_avg_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_scan_traffic.json")
        if world == 1 and os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("n_codes") == n_total and tj.get("bits") == args.bits:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "queries/sec (k-NN top-100) on 128-bit codes, 1B DB; bit-exact vs linear_search",
            "value": Q * args.steps / elapsed,
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[%d]: %d-bit codes, %.3g-code DB%s, top-%d k-NN, linear verify kernel"
                            % (2 if world == 1 else 3, args.bits, n_total,
                               "" if world == 1 else " sharded %d ways by id range" % world, k),
                "n_codes": n_total, "bits": args.bits, "k": k, "queries_per_step": Q, "query_tile": Q,
                "query_kind": "uniform random", "seed": args.seed,
                "parallelism": "1 process/GPU, DB shard per GPU, RCCL all-gather of per-shard top-k + merge kernel"
                               if world > 1 else "single GPU",
                "exchange": exchange,
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                "kernel": "vc_scan_kernel", "launches": tm.scan_launches, "avg_launch_ms": scan_avg_ms,
                "algorithmic_bytes_per_launch": bytes_per_launch,
            },
            "results_check": "ok" if ok else "FAILED",
        }
        if world == 1 and args.cpu_seconds > 0:
            line["cpu_baseline"] = cpu_baseline(args, n_total)
        json_out.write(json.dumps(line) + "\n")
        json_out.flush()
    ss.close()
    if world > 1:
        dist.destroy_process_group()
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
