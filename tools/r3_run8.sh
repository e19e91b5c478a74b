#!/bin/bash
# Dev helper (GPU box): round-3 experiment 8 -- scalar prologue + in-kernel merge without agent fences
cd ${GRAFT_REPO_ROOT:-.}
o=gpurun_out/r3h; mkdir -p $o
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $o/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $o/pytest.log
[ $rc -eq 0 ] || exit $rc
S="--docs 1250000 --emulate-world 8"
bash tools/abl_libs.sh "libsparse_rx.so" "--no-cpu-baseline" "$S --no-cpu-baseline" "$S --no-cpu-baseline --force-dist" "$S --no-cpu-baseline --force-dist --no-overlap" "$S --force-dist" "--workload c2 --no-cpu-baseline" "--workload c1 --no-cpu-baseline" > $o/abl.log 2>&1; cat $o/abl.log
timeout -k 10 300 python tools/stamp_run.py $S > $o/stamp_shard.log 2>&1; tail -14 $o/stamp_shard.log
grep -h "steady state\|host submit" $o/*.log | head
