#!/bin/bash
# Dev helper (GPU box): round-3 run 17 -- dense INT8 filter kernel with the query fragments staged through LDS
cd ${GRAFT_REPO_ROOT:-.}
o=gpurun_out/r3q; mkdir -p $o
timeout -k 10 600 python -m pytest tests/test_dense_int8.py -x -q -m gpu > $o/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $o/pytest.log
[ $rc -eq 0 ] || exit $rc
for a in "1000000 384 1024 100" "1000000 768 1024 10" "1000000 768 1024 100" "4000000 768 1024 100" "1000000 128 1024 100"; do
  timeout -k 10 300 python tools/bench_dense.py $a 2>&1 | grep -v Warning | head -2
done | tee $o/bench_dense.log
