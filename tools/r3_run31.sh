#!/bin/bash
# Dev helper (GPU box): round-3 run 31 -- final dense INT8 evidence: full GPU suite, bench_dense log, rocprofv3 kernel stats
cd ${GRAFT_REPO_ROOT:-.}
o=gpurun_out/r3z7; mkdir -p $o
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $o/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $o/pytest.log
[ $rc -eq 0 ] || exit $rc
for a in "1000000 384 1024 100" "1000000 768 1024 10" "1000000 768 1024 100" "1000000 768 1024 1000" "4000000 768 1024 100" "1000000 128 1024 100" "1000000 1024 1024 100" "1000000 768 64 100"; do
  timeout -k 10 300 python tools/bench_dense.py $a 2>&1 | grep "^dense\|verified"
done > $o/bench_dense.log 2>&1; cat $o/bench_dense.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/trace768 -- python3 tools/bench_dense.py 1000000 768 1024 100 > $o/trace768.log 2>&1; echo "trace rc=$?"
f=$(find $o/trace768 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $o/dense768_kernel_stats.csv
find $o/trace768 -name '*.csv' -size +1M -delete
ls $o
