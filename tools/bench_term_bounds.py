"""Dev bench: the per-term bound table of a C3-sized index (1e9 postings, 100 k terms, Zipf-like run lengths) by the first form
(64-bit torch.sort of (term, value) keys of all postings) and by srx_build_term_bounds.  python tools/bench_term_bounds.py [nnz] [vocab]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sparse_rx
from sparse_rx.index import DeviceIndex

nnz = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
V = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(3)
w = 1.0 / torch.arange(1, V + 1, device=dev, dtype=torch.float64)
df = torch.floor(w / w.sum() * nnz).to(torch.int64)
df[0] += nnz - int(df.sum())
term_ptr = torch.zeros(V + 1, dtype=torch.int64, device=dev); term_ptr[1:] = torch.cumsum(df, 0)
cols_sorted = torch.repeat_interleave(torch.arange(V, device=dev, dtype=torch.int32), df)
post_val = torch.rand(nnz, generator=g, device=dev)
ks = DeviceIndex.FINE_KS


def by_sort():
    bits = post_val.view(torch.int32).to(torch.int64)
    key = (cols_sorted.to(torch.int64) << 32) | (0xFFFFFFFF - bits)
    del bits
    key = torch.sort(key).values
    out = torch.zeros((V, len(ks)), dtype=torch.float32, device=dev)
    for j, K in enumerate(ks):
        has = df >= K
        pos = (term_ptr[:-1] + (K - 1)).clamp(max=nnz - 1)
        b = (0xFFFFFFFF - (key[pos] & 0xFFFFFFFF)).to(torch.int32).view(torch.float32)
        out[:, j] = torch.where(has, b, torch.zeros_like(b))
    return out


def timed(f):
    torch.cuda.synchronize(); torch.cuda.reset_peak_memory_stats(); m0 = torch.cuda.memory_allocated()
    t = time.perf_counter(); r = f(); torch.cuda.synchronize()
    return r, time.perf_counter() - t, (torch.cuda.max_memory_allocated() - m0) / 2**30


a, ta, ma = timed(by_sort)
b, tb, mb = timed(lambda: DeviceIndex._term_bounds(torch, cols_sorted, post_val, term_ptr, df, V))
b, tb, mb = timed(lambda: DeviceIndex._term_bounds(torch, cols_sorted, post_val, term_ptr, df, V))
print(f"term bounds of {nnz} postings, {V} terms, {len(ks)} ranks: 64-bit sort {ta:.3f} s (+{ma:.1f} GiB peak), srx_build_term_bounds {tb:.3f} s (+{mb:.2f} GiB peak); "
      f"equal: {bool(torch.equal(a.view(torch.int32), b.view(torch.int32)))}")
