#!/bin/bash
# Dev helper (GPU box): round-3 experiment 9 -- graph lanes for the sharded step; traces
cd ${GRAFT_REPO_ROOT:-.}
o=gpurun_out/r3i; mkdir -p $o
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "sharded or overlap or edge" > $o/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $o/pytest.log
[ $rc -eq 0 ] || exit $rc
S="--docs 1250000 --emulate-world 8"
for v in "--force-dist" "--force-dist --no-graph" "--force-dist --exchange allgather" ""; do
  timeout -k 10 300 python bench.py $S $v > $o/bench_$(echo $v | tr -d ' -').log 2>&1; echo "[$v] rc=$?"; grep -h "host submit\|PARITY\|capture" $o/bench_$(echo $v | tr -d ' -').log | cut -c1-200
  python - <<PY
import json
for line in open("$o/bench_$(echo $v | tr -d ' -').log"):
    if line.startswith("{"):
        j = json.loads(line); r = j["roofline"]
        print("   step_ms=%.4f wave_ms=%.4f t2=%.4f merge=%.4f submission=%s checked=%s" % (j["ms_per_step"], r["kernel_ms"], r["tier2_kernel_ms"], r["merge_kernel_ms"], j["config"].get("step_submission"), (j.get("parity_check") or {}).get("queries")))
PY
done
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $o/trace_g -- python3 bench.py $S --no-cpu-baseline --force-dist --steps 10 --warmup 3 > $o/trace_g.log 2>&1
f=$(find $o/trace_g -name "*kernel_trace.csv" | head -1)
python3 tools/trace_timeline.py $f 130 > $o/timeline_graph.txt 2>&1
rm -rf $o/trace_g
head -60 $o/timeline_graph.txt
