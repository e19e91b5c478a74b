"""Dev helper: run bench.py against another build of the engine (python tools/bench_with_lib.py <lib.so> <bench args...>)."""
import os, sys, runpy
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import sparse_rx
from sparse_rx import _capi
_capi.LIB_PATH = os.path.join(os.path.dirname(_capi.LIB_PATH), sys.argv[1])
sys.argv = ["bench.py"] + sys.argv[2:]
runpy.run_path(os.path.join(root, "bench.py"), run_name="__main__")
