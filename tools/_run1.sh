cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2j
timeout -k 10 700 python -m pytest tests -x -q -m gpu > gpurun_out/r2j/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/r2j/pytest.log; [ $rc -eq 0 ] || exit $rc
bash tools/abl_libs.sh "libsparse_rx.so" "--workload c4 --steps 5" "--workload c5 --steps 5" "--workload c1" "--workload c2" > gpurun_out/r2j/abl_c4e.log 2>&1; cat gpurun_out/r2j/abl_c4e.log
