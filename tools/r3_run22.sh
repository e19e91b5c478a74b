#!/bin/bash
# Dev helper (GPU box): round-3 run 22 -- dense INT8: eight-wave workgroups on long rows, group-level screen branch
cd ${GRAFT_REPO_ROOT:-.}
o=gpurun_out/r3v; mkdir -p $o
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_dense_int8.py -x -q -m gpu > $o/pytest_dense.log 2>&1; rc=$?; echo "pytest dense rc=$rc"; tail -3 $o/pytest_dense.log
[ $rc -eq 0 ] || exit $rc
for a in "1000000 384 1024 100" "1000000 768 1024 10" "1000000 768 1024 100" "4000000 768 1024 100" "1000000 128 1024 100" "1000000 1024 1024 100" "1000000 768 1024 1000"; do
  timeout -k 10 300 python tools/bench_dense.py $a 2>&1 | grep "^dense int8\|verified" 
done > $o/bench_dense.log 2>&1; cat $o/bench_dense.log
timeout -k 10 300 rocprofv3 --kernel-trace -d $o/dense768 -o dense -- python tools/bench_dense.py 1000000 768 1024 100 > $o/prof768.log 2>&1; echo "prof768 rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace -d $o/dense128 -o dense -- python tools/bench_dense.py 1000000 128 1024 100 > $o/prof128.log 2>&1; echo "prof128 rc=$?"
ls -la $o/dense768 $o/dense128
