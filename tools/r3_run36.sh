#!/bin/bash
# Dev helper (GPU box): round-3 run 36 -- final check of the committed tree: full GPU suite, smoke, default bench
cd ${GRAFT_REPO_ROOT:-.}
o=gpurun_out/r3zc; mkdir -p $o
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $o/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $o/pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $o/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $o/smoke.log
timeout -k 10 300 python bench.py > $o/bench_c3.log 2>&1; echo "[c3] rc=$?"; tail -1 $o/bench_c3.log | cut -c1-300
