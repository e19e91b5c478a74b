#!/bin/bash
# Dev helper (GPU box): round-3 run 13 -- tier 2 templated on AFTER: tests, C4 / C5, where a C4 block's time goes
cd ${GRAFT_REPO_ROOT:-.}
o=gpurun_out/r3m; mkdir -p $o
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $o/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $o/pytest.log
[ $rc -eq 0 ] || exit $rc
bash tools/abl_libs.sh "libsparse_rx.so" "--workload c4 --no-cpu-baseline --steps 10" "--workload c5 --no-cpu-baseline --steps 10" "--workload c1 --no-cpu-baseline" > $o/abl.log 2>&1; cat $o/abl.log
timeout -k 10 300 python tools/stamp2_run.py --workload c4 > $o/stamp2_c4.log 2>&1; tail -9 $o/stamp2_c4.log
timeout -k 10 300 python tools/stamp2_run.py --workload c5 > $o/stamp2_c5.log 2>&1; tail -9 $o/stamp2_c5.log
