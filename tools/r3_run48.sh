#!/bin/bash
# Dev helper (GPU box): round-3 run 48 -- dense tests + bench log on the final dense kernel (non-temporal B loads)
cd ${GRAFT_REPO_ROOT:-.}
o=gpurun_out/r3zq; mkdir -p $o
timeout -k 10 600 python -m pytest tests/test_dense_int8.py -x -q -m gpu > $o/pytest_dense.log 2>&1; rc=$?; echo "pytest dense rc=$rc"; tail -3 $o/pytest_dense.log
[ $rc -eq 0 ] || exit $rc
for mode in packed rowmajor; do
  for a in "1000000 384 1024 100" "1000000 768 1024 10" "1000000 768 1024 100" "1000000 768 1024 1000" "4000000 768 1024 100" "1000000 128 1024 100" "1000000 1024 1024 100" "1000000 1024 1024 10" "1000000 768 64 100" "1000000 768 256 100"; do
    echo "== $mode"; if [ $mode = rowmajor ]; then export SRX_DENSE_ROWMAJOR=1; else unset SRX_DENSE_ROWMAJOR; fi
    timeout -k 10 300 python tools/bench_dense.py $a 2>&1 | grep "^dense\|verified"
  done
done > $o/bench_dense.log 2>&1; grep "^dense int8\|^==" $o/bench_dense.log | paste - - | cut -c1-140
