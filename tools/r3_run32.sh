#!/bin/bash
# Dev helper (GPU box): round-3 run 32 -- srx_build_term_bounds: test, full suite, build times of C3 / C4 / C5
cd ${GRAFT_REPO_ROOT:-.}
o=gpurun_out/r3z8; mkdir -p $o
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "term_bounds or blocked_layout" > $o/pytest_tb.log 2>&1; rc=$?; echo "pytest tb rc=$rc"; tail -3 $o/pytest_tb.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $o/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $o/pytest.log
[ $rc -eq 0 ] || exit $rc
for w in c3 c4 c5; do
  timeout -k 10 400 python bench.py --workload $w --no-cpu-baseline > $o/bench_$w.log 2>&1; echo "[$w] rc=$?"; tail -1 $o/bench_$w.log | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"index_build_s": [0-9.]*'
done
