cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/pmcq; mkdir -p gpurun_out/pmcq
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pmcq/a -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmcq/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/pmcq/b -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmcq/b.log 2>&1
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmcq/*/**/*counter_collection.csv", recursive=True)):
    acc=collections.defaultdict(lambda:[0.0,0])
    for row in csv.DictReader(open(f)):
        if "srx_wave_kernel" in row.get("Kernel_Name",""):
            acc[row["Counter_Name"]][0]+=float(row["Counter_Value"]); acc[row["Counter_Name"]][1]+=1
    for k,(s,n) in sorted(acc.items()): print(f"{k:26s} {s/n:.5g}")
PY
