#!/bin/bash
# Dev helper (GPU box): round-3 run 43 -- dense INT8: smaller threshold sample for small k
cd ${GRAFT_REPO_ROOT:-.}
o=gpurun_out/r3zk; mkdir -p $o
timeout -k 10 600 python -m pytest tests/test_dense_int8.py -x -q -m gpu > $o/pytest_dense.log 2>&1; rc=$?; echo "pytest dense rc=$rc"; tail -3 $o/pytest_dense.log
[ $rc -eq 0 ] || exit $rc
for a in "1000000 768 1024 10" "1000000 768 1024 25" "1000000 768 1024 50" "1000000 768 1024 100" "1000000 384 1024 10" "1000000 1024 1024 10" "4000000 768 1024 10" "1000000 768 64 10"; do
  timeout -k 10 300 python tools/bench_dense.py $a 2>&1 | grep "^dense int8\|verified"
done > $o/bench_dense.log 2>&1; grep -v verified $o/bench_dense.log; grep -c verified $o/bench_dense.log
