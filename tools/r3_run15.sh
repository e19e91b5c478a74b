#!/bin/bash
# Dev helper (GPU box): round-3 run 15 -- one copy of the postings with the specialised dense paths: tests, with / without the canonical copy
cd ${GRAFT_REPO_ROOT:-.}
o=gpurun_out/r3o; mkdir -p $o
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $o/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $o/pytest.log
[ $rc -eq 0 ] || exit $rc
bash tools/abl_libs.sh "libsparse_rx.so" "--workload c4 --steps 10" "--workload c4 --steps 10 --drop-canonical" "--workload c5 --steps 10" "--workload c5 --steps 10 --drop-canonical" "--workload c1" "--workload c1 --drop-canonical" "--no-cpu-baseline" "--no-cpu-baseline --drop-canonical" > $o/abl.log 2>&1; cat $o/abl.log
grep -h "device_index_mb" $o/*.log | head -2
