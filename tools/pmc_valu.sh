cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
# usage: bash tools/pmc_valu.sh <tag> "<bench args 1>" "<bench args 2>" ...: one PMC pass (instruction counts of the tier-1
# kernel) per argument set; per-dispatch averages to gpurun_out/pmcv_<tag>.txt
tag=$1; shift
rm -rf gpurun_out/pmcv_$tag; mkdir -p gpurun_out/pmcv_$tag
i=0
for a in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pmcv_$tag/$i -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline $a > gpurun_out/pmcv_$tag/$i.log 2>&1
  echo "== [$a]" >> gpurun_out/pmcv_$tag.txt
  python3 - <<PY >> gpurun_out/pmcv_$tag.txt
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmcv_$tag/$i/**/*counter_collection.csv", recursive=True)):
    acc=collections.defaultdict(lambda:[0.0,0])
    for row in csv.DictReader(open(f)):
        if "srx_wave_kernel" in row.get("Kernel_Name",""):
            acc[row["Counter_Name"]][0]+=float(row["Counter_Value"]); acc[row["Counter_Name"]][1]+=1
    print("  ".join(f"{k}={s/n:.5g}" for k,(s,n) in sorted(acc.items())))
PY
done
cat gpurun_out/pmcv_$tag.txt
