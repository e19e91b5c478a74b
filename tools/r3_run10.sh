#!/bin/bash
# Dev helper (GPU box): round-3 run 10 -- tear-down of the graph lanes, exchange timeline, final bench logs
cd ${GRAFT_REPO_ROOT:-.}
o=gpurun_out/r3j; mkdir -p $o
S="--docs 1250000 --emulate-world 8"
for v in "--force-dist" "--force-dist --exchange allgather" "--force-dist --no-graph"; do
  n=$(echo $v | tr -d ' -')
  timeout -k 10 200 python bench.py $S $v > $o/bench_shard_$n.log 2>&1; echo "[$v] rc=$?"; grep -h "host submit\|PARITY\|capture" $o/bench_shard_$n.log | cut -c1-200
done
timeout -k 10 200 python bench.py $S > $o/bench_shard_emulate8.log 2>&1; echo "[shard] rc=$?"
export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $o/trace_g -- python3 bench.py $S --no-cpu-baseline --force-dist --steps 10 --warmup 3 > $o/trace_g.log 2>&1; echo "trace rc=$?"
f=$(find $o/trace_g -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python3 tools/trace_timeline.py $f 130 > $o/timeline_graph.txt 2>&1
rm -rf $o/trace_g
head -50 $o/timeline_graph.txt
for w in c3 c2 c1 c5 c4; do
  timeout -k 10 400 python bench.py --workload $w > $o/bench_$w.log 2>&1; echo "[$w] rc=$?"; tail -1 $o/bench_$w.log | cut -c1-260
done
timeout -k 10 300 python bench.py --gpus 1 --self-launch --workload c2 > $o/bench_selflaunch_c2.log 2>&1; echo "[self-launch] rc=$?"
