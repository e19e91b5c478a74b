#!/bin/bash
# Dev helper (GPU box): round-3 final evidence -- tests, bench logs of every workload, shard rehearsals, rocprofv3 + PMC of C3 / C5 / C4
cd ${GRAFT_REPO_ROOT:-.}
o=gpurun_out/r3z; mkdir -p $o
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $o/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $o/pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $o/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $o/smoke.log
for w in c3 c2 c1 c5 c4; do
  timeout -k 10 400 python bench.py --workload $w > $o/bench_$w.log 2>&1; echo "[$w] rc=$?"; tail -1 $o/bench_$w.log | cut -c1-200
done
S="--docs 1250000 --emulate-world 8"
timeout -k 10 200 python bench.py $S > $o/bench_shard_emulate8.log 2>&1; echo "[shard] rc=$?"
timeout -k 10 200 python bench.py $S --force-dist > $o/bench_shard_dist_a2a_graph.log 2>&1; echo "[shard dist] rc=$?"
timeout -k 10 200 python bench.py $S --force-dist --exchange allgather > $o/bench_shard_dist_allgather_graph.log 2>&1; echo "[shard dist ag] rc=$?"
timeout -k 10 200 python bench.py $S --force-dist --no-graph > $o/bench_shard_dist_a2a_eager.log 2>&1; echo "[shard dist eager] rc=$?"
bash tools/gpu_profile.sh c3r3 > $o/profile_c3.log 2>&1; echo "profile c3 rc=$?"
TRACE_STEPS=10 bash tools/gpu_profile.sh c5r3 --workload c5 > $o/profile_c5.log 2>&1; echo "profile c5 rc=$?"
TRACE_STEPS=10 bash tools/gpu_profile.sh c4r3 --workload c4 > $o/profile_c4.log 2>&1; echo "profile c4 rc=$?"
