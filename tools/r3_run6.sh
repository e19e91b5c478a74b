#!/bin/bash
# Dev helper (GPU box): round-3 experiment 6 -- where a tier-1 wave's time goes (stamps), exchange overlap with own streams
cd ${GRAFT_REPO_ROOT:-.}
o=gpurun_out/r3f; mkdir -p $o
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $o/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $o/pytest.log
[ $rc -eq 0 ] || exit $rc
S="--docs 1250000 --emulate-world 8"
timeout -k 10 300 python tools/stamp_run.py $S > $o/stamp_shard.log 2>&1; tail -16 $o/stamp_shard.log
timeout -k 10 300 python tools/stamp_run.py > $o/stamp_c3.log 2>&1; tail -16 $o/stamp_c3.log
bash tools/abl_libs.sh "libsparse_rx.so" "--no-cpu-baseline" "$S --no-cpu-baseline" "$S --no-cpu-baseline --force-dist" "$S --no-cpu-baseline --force-dist --no-overlap" "$S --no-cpu-baseline --force-dist --exchange allgather" "--workload c2 --no-cpu-baseline" "--workload c1 --no-cpu-baseline" > $o/abl.log 2>&1; cat $o/abl.log
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $o/trace_ov -- python3 bench.py $S --no-cpu-baseline --force-dist --steps 10 --warmup 3 > $o/trace_ov.log 2>&1
f=$(find $o/trace_ov -name "*kernel_trace.csv" | head -1)
python3 tools/trace_timeline.py $f 90 > $o/timeline_ov.txt 2>&1
rm -rf $o/trace_ov
head -40 $o/timeline_ov.txt
grep -h "steady state" $o/*.log | head
