#!/bin/bash
# Dev helper (GPU box): round-3 run 11 -- tests after the store-wait fix, C2 steady-state legs, rocprofv3 + PMC of C3
cd ${GRAFT_REPO_ROOT:-.}
o=gpurun_out/r3k; mkdir -p $o
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $o/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $o/pytest.log
[ $rc -eq 0 ] || exit $rc
for i in 1 2 3; do timeout -k 10 300 python bench.py --workload c2 > $o/bench_c2_$i.log 2>&1; echo "[c2 #$i] rc=$?"; grep -h "steady state\|PARITY" $o/bench_c2_$i.log | cut -c1-250; done
timeout -k 10 300 python bench.py --workload c1 --streams 4 > $o/bench_c1.log 2>&1; echo "[c1] rc=$?"; grep -h "steady state\|PARITY" $o/bench_c1.log | cut -c1-250
bash tools/gpu_profile.sh c3r3 > $o/profile_c3.log 2>&1; echo "profile rc=$?"; tail -30 $o/profile_c3.log | cut -c1-200
