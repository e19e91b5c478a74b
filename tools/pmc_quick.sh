cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
# usage: bash tools/pmc_quick.sh <tag> [bench args]: two PMC passes over the tier-1 kernel, per-dispatch averages to gpurun_out/pmcq_<tag>.txt
tag=$1; shift
rm -rf gpurun_out/pmcq_$tag; mkdir -p gpurun_out/pmcq_$tag
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pmcq_$tag/a -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/pmcq_$tag/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/pmcq_$tag/b -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/pmcq_$tag/b.log 2>&1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_IFETCH SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_MISC --output-format csv -d gpurun_out/pmcq_$tag/c -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/pmcq_$tag/c.log 2>&1
python3 - <<PY > gpurun_out/pmcq_$tag.txt
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmcq_$tag/*/**/*counter_collection.csv", recursive=True)):
    acc=collections.defaultdict(lambda:[0.0,0])
    for row in csv.DictReader(open(f)):
        if "srx_wave_kernel" in row.get("Kernel_Name",""):
            acc[row["Counter_Name"]][0]+=float(row["Counter_Value"]); acc[row["Counter_Name"]][1]+=1
    for k,(s,n) in sorted(acc.items()): print(f"{k:26s} {s/n:.5g}")
PY
cat gpurun_out/pmcq_$tag.txt
