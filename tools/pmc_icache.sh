cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/pmci; mkdir -p gpurun_out/pmci
rocprofv3 -L 2>/dev/null | grep -io "SQC_[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_INST[A-Z_]*\|SQ_INST_CYCLES[A-Z_]*" | sort -u | tr '\n' ' ' > gpurun_out/pmci/names.txt
cat gpurun_out/pmci/names.txt; echo
LIB=${1:-libsparse_rx_v12.so}
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_IFETCH --output-format csv -d gpurun_out/pmci/a -- python3 tools/bench_with_lib.py $LIB --steps 3 --warmup 1 --no-cpu-baseline --docs 1250000 > gpurun_out/pmci/a.log 2>&1
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_IFETCH --output-format csv -d gpurun_out/pmci/b -- python3 tools/bench_with_lib.py $LIB --steps 3 --warmup 1 --no-cpu-baseline --docs 1250000 --debug 2 > gpurun_out/pmci/b.log 2>&1
tail -3 gpurun_out/pmci/a.log | cut -c1-300
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmci/*/**/*counter_collection.csv", recursive=True)):
    print(f)
    acc=collections.defaultdict(lambda:[0.0,0])
    for row in csv.DictReader(open(f)):
        if "srx_wave_kernel" in row.get("Kernel_Name",""):
            acc[row["Counter_Name"]][0]+=float(row["Counter_Value"]); acc[row["Counter_Name"]][1]+=1
    for k,(s,n) in sorted(acc.items()): print(f"{k:26s} {s/n:.5g}")
PY
