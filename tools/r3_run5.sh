#!/bin/bash
# Dev helper (GPU box): round-3 experiment 5 -- 5 waves / SIMD tier 1, event overhead, 2-D layout rehearsal, exchange trace
cd ${GRAFT_REPO_ROOT:-.}
o=gpurun_out/r3e; mkdir -p $o
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $o/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $o/pytest.log
[ $rc -eq 0 ] || exit $rc
S="--docs 1250000 --emulate-world 8 --no-cpu-baseline"
bash tools/abl_libs.sh "libsparse_rx_w5d2.so libsparse_rx_w4d2.so libsparse_rx_w4d3.so" "--no-cpu-baseline" "$S" "$S --target-blocks 4096" "$S --target-blocks 5120" "--workload c2 --no-cpu-baseline" > $o/abl.log 2>&1; cat $o/abl.log
bash tools/abl_libs.sh "libsparse_rx.so" "$S --profile-every 4" "$S --profile-every 0" "--no-cpu-baseline --profile-every 4" "--docs 5000000 --queries 2500 --emulate-world 2 --no-cpu-baseline" "--docs 2500000 --queries 5000 --emulate-world 4 --no-cpu-baseline" > $o/abl2.log 2>&1; cat $o/abl2.log
export TMPDIR=/tmp
for mode in ov noov; do
  extra=""; [ $mode = noov ] && extra="--no-overlap"
  rocprofv3 --kernel-trace --output-format csv -d $o/trace_$mode -- python3 bench.py $S --force-dist --steps 10 --warmup 3 $extra > $o/trace_$mode.log 2>&1
  f=$(find $o/trace_$mode -name "*kernel_trace.csv" | head -1)
  python3 tools/trace_timeline.py $f 90 > $o/timeline_$mode.txt 2>&1
  rm -rf $o/trace_$mode
  tail -25 $o/timeline_$mode.txt
done
