"""Dev probe: PCIe copy rates of this box for the result rows of a 10 k-query batch (8 MB), the way HostBatchPipeline
moves them (srx_memcpy_async on a copy stream into pinned memory), against torch's own copy."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sparse_rx import _capi
L = _capi.lib()
dev = torch.device("cuda:0")
for mb in (1, 8, 64):
    n = mb * (1 << 20) // 4
    d = torch.zeros(n, dtype=torch.int32, device=dev)
    h = torch.empty(n, dtype=torch.int32).pin_memory()
    s = torch.cuda.Stream(device=dev)
    for name, fn in (("srx_memcpy_async D2H", lambda: L.srx_memcpy_async(h.data_ptr(), d.data_ptr(), 4 * n, s.cuda_stream)),
                     ("srx_memcpy_async H2D", lambda: L.srx_memcpy_async(d.data_ptr(), h.data_ptr(), 4 * n, s.cuda_stream))):
        for _ in range(3):
            fn()
        s.synchronize()
        t = time.perf_counter()
        for _ in range(20):
            fn()
        s.synchronize()
        dt = (time.perf_counter() - t) / 20
        print(f"{mb:3d} MB {name}: {1e3 * dt:.3f} ms  = {mb / 1024 / dt:.1f} GB/s")
    with torch.cuda.stream(s):
        for _ in range(3):
            h.copy_(d, non_blocking=True)
        s.synchronize()
        t = time.perf_counter()
        for _ in range(20):
            h.copy_(d, non_blocking=True)
        s.synchronize()
        dt = (time.perf_counter() - t) / 20
        print(f"{mb:3d} MB torch copy_ D2H      : {1e3 * dt:.3f} ms  = {mb / 1024 / dt:.1f} GB/s")
