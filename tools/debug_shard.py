"""Dev tool: two ranks on ONE GPU (gloo): is the per-shard result right, is the gathered buffer right?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
from verticut_amd import engine as vc
from verticut_amd.sharded import ShardedSearch
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
n, bits, k, Q = int(2e8), 128, 100, 8
ss = ShardedSearch(bits, n, rank=rank, world=world, device=0, query_tile=Q, pipelined=False)
ss.add_synthetic(34)
rng = np.random.default_rng(35)
batches = [rng.integers(0, 256, size=(Q, 16), dtype=np.uint8) for _ in range(4)]
dev = [torch.from_numpy(b).cuda() for b in batches]
for i in range(6):
    out, cnt = ss.search(dev[i % 4], k)
    torch.cuda.synchronize()
    merged = out.cpu().numpy().view(np.uint64).copy()
    local_async = ss._local.cpu().numpy().view(np.uint64).copy()
    gath = ss._gath.cpu().numpy().view(np.uint64).copy()
    local_host, _ = ss.backend.engine.search_knn(batches[i % 4], k)      # host API, synchronous, same engine
    # every rank's local list via a plain CPU all_gather of the host-API result
    lists = [None] * world
    dist.all_gather_object(lists, local_host)
    ref = np.sort(np.concatenate(lists, axis=1), axis=1)[:, :k]
    print(f"rank {rank} step {i}: local_dev==local_host {np.array_equal(local_async, local_host)}  "
          f"gath[own]==local {np.array_equal(gath[rank], local_host)}  gath[other]==other's {np.array_equal(gath[1-rank], lists[1-rank])}  "
          f"merged==ref {np.array_equal(merged, ref)}", flush=True)
dist.destroy_process_group()
