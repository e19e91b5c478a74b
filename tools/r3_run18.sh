#!/bin/bash
# Dev helper (GPU box): round-3 run 18 -- per-kernel trace of the dense INT8 path; full GPU suite and shard rehearsals after the
# per-lane communicator change
cd ${GRAFT_REPO_ROOT:-.}
o=gpurun_out/r3r; mkdir -p $o
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $o/dense768 -o dense -- python tools/bench_dense.py 1000000 768 1024 100 > $o/prof768.log 2>&1; echo "prof768 rc=$?"
rocprofv3 --kernel-trace --stats -d $o/dense128 -o dense -- python tools/bench_dense.py 1000000 128 1024 100 > $o/prof128.log 2>&1; echo "prof128 rc=$?"
for d in dense768 dense128; do f=$(find $o/$d -name '*kernel_stats.csv' | head -1); echo "== $d"; head -12 $f | cut -c1-200; done
find $o -name '*.csv' ! -name '*kernel_stats.csv' -delete; find $o -name '*.db' -delete
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $o/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $o/pytest.log
[ $rc -eq 0 ] || exit $rc
S="--docs 1250000 --emulate-world 8"
timeout -k 10 200 python bench.py $S --force-dist > $o/bench_shard_dist_a2a_graph.log 2>&1; echo "[shard dist] rc=$?"; tail -1 $o/bench_shard_dist_a2a_graph.log | cut -c1-160
timeout -k 10 200 python bench.py $S --force-dist --exchange allgather > $o/bench_shard_dist_allgather_graph.log 2>&1; echo "[shard dist ag] rc=$?"; tail -1 $o/bench_shard_dist_allgather_graph.log | cut -c1-160
