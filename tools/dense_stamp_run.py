"""Diagnostic: run one dense INT8 shape against the -DSRX_DSTAMP build (bash tools/build_variant.sh dstamp -DSRX_DSTAMP) and print
where a filter-kernel wave's time goes.  python tools/dense_stamp_run.py [n_docs] [dim] [nq] [k]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sparse_rx
from sparse_rx import _capi
_capi.LIB_PATH = os.path.join(os.path.dirname(_capi.LIB_PATH), os.environ.get("SRX_LIB", "libsparse_rx_dstamp.so"))
n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 768
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
k = int(sys.argv[4]) if len(sys.argv) > 4 else 100
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
c = torch.randint(-127, 128, (n_docs, dim), generator=g, device=dev, dtype=torch.int32).to(torch.int8)
cs = torch.rand(n_docs, generator=g, device=dev) + 0.01
q = torch.randint(-127, 128, (nq, dim), generator=g, device=dev, dtype=torch.int32).to(torch.int8)
qs = (torch.rand(nq, generator=g, device=dev) + 0.01) / 127
ix = sparse_rx.DenseInt8Index(c, cs)
L = _capi.lib()
L.srx_debug_read_dstamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
out = (ctypes.c_ulonglong * 16)()
for _ in range(2):
    ix.search_device(q, qs, k)
torch.cuda.synchronize()
L.srx_debug_read_dstamps(out)
steps = 3
for _ in range(steps):
    ix.search_device(q, qs, k)
torch.cuda.synchronize()
L.srx_debug_read_dstamps(out)
names = ["prologue (B fragments, first tile)", "stage issue", "MFMA loop (issue)", "epilogue", "flush / table / wait staged tile", "barrier", "final flush"]
w = max(out[8], 1)
tot = sum(out[i] for i in range(7))
print(f"{n_docs} x {dim}, {nq} queries, k={k}: waves {w // steps} per search, {tot / w / 100.0:.1f} us per wave (100 MHz ticks)")
for i, n in enumerate(names):
    print(f"{n:36s} {100.0 * out[i] / tot:6.2f} %   {out[i] / w / 100.0:9.2f} us/wave")
