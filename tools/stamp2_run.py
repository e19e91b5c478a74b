"""Diagnostic: run a bench workload against the SRX_STAMP2 build and print where a tier-2 block's time goes (thread 0's clock)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparse_rx
from sparse_rx import _capi
_capi.LIB_PATH = os.path.join(os.path.dirname(_capi.LIB_PATH), "libsparse_rx_stamp2.so")
import runpy
sys.argv = ["bench.py", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"] + sys.argv[1:]
try:
    runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), run_name="__main__")
except SystemExit:
    pass
L = ctypes.CDLL(_capi.LIB_PATH)
out = (ctypes.c_ulonglong * 16)()
L.srx_debug_read_stamps2.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
L.srx_debug_read_stamps2(out)
names = ["between phases (setup, skip rows, packer)", "flat tiles", "hash units", "dense accumulate", "dense select", "final shrink"]
tot = sum(out[i] for i in range(6))
print(f"calls: flat {out[9]}  hash {out[10]}  dense {out[11]}  blocks {out[12]}  compact selects {out[13]}  general selects after roll-back {out[14]}")
for i, n in enumerate(names):
    cnt = {1: out[9], 2: out[10], 3: out[11], 4: out[11], 5: out[12]}.get(i, out[12])
    print(f"{n:44s} {100.0 * out[i] / max(tot, 1):6.2f} %   {out[i] / max(cnt, 1):12.0f} ticks/call")
print(f"inside dense select: scan+append {out[6] / max(out[11], 1):.0f} ticks/tile-group, compact selects {out[7] / max(out[13], 1):.0f} ticks each ({100.0 * out[7] / max(tot, 1):.2f} % of all)")
