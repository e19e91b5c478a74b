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
out = (ctypes.c_ulonglong * 32)()
L.srx_debug_read_stamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
L.srx_debug_read_stamps(out)
names = ["loop / previous tail", "issue", "wait data + pass 1", "multi-term docs", "restore", "screening", "candidates", "final select",
         "prologue", "restart selections", "rank + row write", "-"]
tot = sum(out[i] for i in range(12))
w = max(out[12], 1)
print("waves:", out[12], "ticks/wave:", tot / w, "(100 MHz ticks: %.1f us per wave)" % (tot / w / 100.0))
c = [out[13 + i] / w for i in range(8)]
print(f"per wave: units {c[0]:.1f}  units with multi-term docs {c[1]:.1f}  docs resolved {c[2]:.1f}  screening triggers {c[3]:.1f}  restarts {c[4]:.2f}  single-term candidates appended {c[5]:.1f}")
for i, n in enumerate(names):
    print(f"{n:28s} {100.0 * out[i] / tot:6.2f} %   {out[i] / w:12.0f} ticks/wave")
