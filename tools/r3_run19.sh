#!/bin/bash
# Dev helper (GPU box): round-3 run 19 -- dense INT8 filter with the fp32 screen + per-wave survivor lists; full suite; shard rehearsals
cd ${GRAFT_REPO_ROOT:-.}
o=gpurun_out/r3s; mkdir -p $o
timeout -k 10 600 python -m pytest tests/test_dense_int8.py -x -q -m gpu > $o/pytest_dense.log 2>&1; rc=$?; echo "pytest dense rc=$rc"; tail -3 $o/pytest_dense.log
[ $rc -eq 0 ] || exit $rc
for a in "1000000 384 1024 100" "1000000 768 1024 10" "1000000 768 1024 100" "4000000 768 1024 100" "1000000 128 1024 100" "1000000 1024 1024 100"; do
  timeout -k 10 300 python tools/bench_dense.py $a 2>&1 | grep "^dense int8\|verified" 
done > $o/bench_dense.log 2>&1; cat $o/bench_dense.log
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $o/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $o/pytest.log
[ $rc -eq 0 ] || exit $rc
S="--docs 1250000 --emulate-world 8"
timeout -k 10 200 python bench.py $S --force-dist > $o/bench_shard_dist_a2a_graph.log 2>&1; echo "[shard dist] rc=$?"; tail -1 $o/bench_shard_dist_a2a_graph.log | cut -c1-160
timeout -k 10 200 python bench.py $S --force-dist --exchange allgather > $o/bench_shard_dist_allgather_graph.log 2>&1; echo "[shard dist ag] rc=$?"; tail -1 $o/bench_shard_dist_allgather_graph.log | cut -c1-160
