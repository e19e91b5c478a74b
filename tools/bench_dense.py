"""Dev bench of the dense INT8 path: random int8 corpus / queries resident in HBM, srx_dense_search_i8,
sample verified against the oracle.  python tools/bench_dense.py [n_docs] [dim] [nq] [k]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sparse_rx
from oracle import np_oracle
if os.environ.get("SRX_LIB"):  # dev: an engine variant of tools/build_variant.sh
    sparse_rx._capi.LIB_PATH = os.path.join(os.path.dirname(sparse_rx._capi.LIB_PATH), os.environ["SRX_LIB"])

n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 384
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
k = int(sys.argv[4]) if len(sys.argv) > 4 else 100
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
c = torch.randint(-127, 128, (n_docs, dim), generator=g, device=dev, dtype=torch.int32).to(torch.int8)
cs = torch.rand(n_docs, generator=g, device=dev) + 0.01
q = torch.randint(-127, 128, (nq, dim), generator=g, device=dev, dtype=torch.int32).to(torch.int8)
qs = (torch.rand(nq, generator=g, device=dev) + 0.01) / 127
ix = sparse_rx.DenseInt8Index(c, cs, packed=not os.environ.get("SRX_DENSE_ROWMAJOR"))  # dev: the row-major corpus for comparison
for _ in range(2):
    out = ix.search_device(q, qs, k)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
steps = 5
a.record()
for _ in range(steps):
    out = ix.search_device(q, qs, k)
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / steps
ops = 2.0 * nq * n_docs * ix.dim_pad
mat = 2.0 * nq * n_docs * 4  # the score matrix is written and read once
print(f"dense int8: {n_docs} docs x {dim} dim, {nq} queries, k={k}: {ms:.3f} ms/batch, {nq / ms * 1e3:,.0f} queries/s, "
      f"{ops / ms / 1e9:.1f} int8 TOP/s, score-matrix traffic {mat / ms / 1e6:.0f} GB/s")
# verify a sample against the oracle
sq = min(nq, 8); sd = min(n_docs, 200_000)
ixs = sparse_rx.DenseInt8Index(c[:sd], cs[:sd])
d, s, n = ixs.search(q[:sq].cpu().numpy(), qs[:sq].cpu().numpy(), k)
ed, es, en = np_oracle.dense_topk(np_oracle.int8_similarities(q[:sq].cpu().numpy(), c[:sd].cpu().numpy(), qs[:sq].cpu().numpy(), cs[:sd].cpu().numpy()), k)
assert np.array_equal(n, en) and np.array_equal(d, ed) and np.array_equal(s.view(np.uint32), es.view(np.uint32))
print("sample verified bit-exact against the oracle")

# ---- f32 search_by_vector (one query at a time is the reference's API; a pass takes up to 4) ----
e = torch.randn((n_docs, dim), generator=g, device=dev)
fx = sparse_rx.DenseF32Index(e)
for nqf in (1, 4):
    qf = torch.randn((nqf, dim), generator=g, device=dev)
    for _ in range(2):
        fx.search_device(qf, k)
    torch.cuda.synchronize()
    a.record()
    for _ in range(10):
        fx.search_device(qf, k)
    b.record(); torch.cuda.synchronize()
    msf = a.elapsed_time(b) / 10
    print(f"dense f32: {n_docs} docs x {dim} dim, {nqf} quer{'y' if nqf == 1 else 'ies'}, k={k}: {msf:.3f} ms, matrix stream {n_docs * fx.dim_pad * 4 / msf / 1e6:.0f} GB/s")
