"""Dev helper: GPU time of each stage of the sharded search's exchange at world size 1 (RCCL process group of one rank)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
import sparse_rx
from sparse_rx import synth
from sparse_rx.index import merge_topk_packed_out_device

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
nq, k, world = 10000, 100, 1
row = 2 * k + 1
W = int(sys.argv[1]) if len(sys.argv) > 1 else 8   # emulate the merge shape of W ranks: nq/W queries x W lists
send = torch.zeros((nq, row), dtype=torch.int32, device=dev)
send[:, 2 * k] = k
send[:, :k] = torch.arange(k, device=dev, dtype=torch.int32)
send[:, k:2 * k] = torch.rand((nq, k), device=dev).sort(dim=1, descending=True).values.view(torch.int32)
recv = torch.empty_like(send)
blk = nq // W
lists = send[: W * blk].view(W, blk, row).contiguous()
merged = torch.empty((blk, row), dtype=torch.int32, device=dev)
allrows = torch.empty((nq, row), dtype=torch.int32, device=dev)

def timeit(name, fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    print(f"{name:40s} {a.elapsed_time(b) / n * 1e3:8.1f} us")

timeit("all_to_all_single 8 MB (world 1)", lambda: dist.all_to_all_single(recv, send))
timeit("all_gather_into_tensor 8 MB (world 1)", lambda: dist.all_gather_into_tensor(allrows, send))
timeit(f"merge_packed_out {blk} queries x {W} lists", lambda: merge_topk_packed_out_device(lists, k, merged))
timeit("merge_packed_out 10000 queries x 1 list", lambda: merge_topk_packed_out_device(send.view(1, nq, row), k, allrows))
timeit("copy 8 MB", lambda: recv.copy_(send))
dist.destroy_process_group()
