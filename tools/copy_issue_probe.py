import sys, time, ctypes
sys.path.insert(0, ".")
import torch, numpy as np
import sparse_rx
from sparse_rx import _capi
L = _capi.lib()
dev = torch.device("cuda:0")
h = torch.empty(2_010_000, dtype=torch.int32).pin_memory(); d = torch.empty(2_010_000, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
def bench(name, f, n=50):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{name:50s} issue {1e3*(t1-t)/n:.3f} ms/call   drained after {1e3*(t2-t1):.3f} ms")
s_def = torch.cuda.current_stream(dev)
s_new = torch.cuda.Stream(device=dev)
s_hi = torch.cuda.Stream(device=dev, priority=-1)
for nm, st in (("default", s_def), ("torch.cuda.Stream()", s_new), ("high-priority stream", s_hi)):
    bench(f"D2H 8 MB srx_memcpy_async on {nm}", lambda: L.srx_memcpy_async(h.data_ptr(), d.data_ptr(), 8_040_000, st.cuda_stream))
    bench(f"H2D 0.7 MB srx_memcpy_async on {nm}", lambda: L.srx_memcpy_async(d.data_ptr(), h.data_ptr(), 700_000, st.cuda_stream))
ev = torch.cuda.Event()
bench("torch Event.record(default)", lambda: ev.record(s_def))
bench("torch Event.record(new stream)", lambda: ev.record(s_new))
bench("stream.wait_event", lambda: s_new.wait_event(ev))
print("h pinned:", h.is_pinned())
