"""Diagnostic: run the C3 bench workload against the SRX_STAMP build and print where a tier-1 wave's cycles go."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparse_rx
from sparse_rx import _capi
_capi.LIB_PATH = os.path.join(os.path.dirname(_capi.LIB_PATH), "libsparse_rx_stamp.so")
import runpy
sys.argv = ["bench.py", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"] + sys.argv[1:]
try:
    runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), run_name="__main__")
except SystemExit:
    pass
L = _capi.lib()
out = (ctypes.c_ulonglong * 16)()
L.srx_debug_read_stamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
L.srx_debug_read_stamps(out)
names = ["loop / previous tail", "issue", "wait data + pass 1", "multi-term docs", "restore", "screening", "candidates", "epilogue"]
tot = sum(out[i] for i in range(8))
print("waves:", out[8], "ticks/wave:", tot / max(out[8], 1))
w = max(out[8], 1)
print(f"per wave: units {out[9]/w:.1f}  units with multi-term docs {out[10]/w:.1f}  docs resolved {out[11]/w:.1f}  screening triggers {out[12]/w:.1f}")
for i, n in enumerate(names):
    print(f"{n:28s} {100.0 * out[i] / tot:6.2f} %   {out[i] / max(out[8],1):12.0f} ticks/wave")
