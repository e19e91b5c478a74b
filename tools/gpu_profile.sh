#!/bin/bash
# Dev helper (GPU box): rocprofv3 kernel trace + PMC passes of bench.py; summaries land in gpurun_out/prof_<tag>/
cd ${GRAFT_REPO_ROOT:-.}
tag=$1; shift
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
echo "$@" > $out/args.txt
export TMPDIR=/tmp
ARGS="bench.py --steps 5 --warmup 1 --no-cpu-baseline $@"
# the kernel-trace pass runs the bench's own default protocol (3 warm-up + 20 timed steps), so that its per-launch
# durations are comparable with the hipEvent figure of a default run (the first launches of a process are slower:
# clocks ramp up)
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps ${TRACE_STEPS:-20} --warmup 3 --no-cpu-baseline $@ > $out/trace.log 2>&1
echo "trace rc=$?" >> $out/summary.txt
for pmc in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD"; do
  name=$(echo $pmc | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $pmc --output-format csv -d $out/pmc_$name -- python3 $ARGS > $out/pmc_$name.log 2>&1
  echo "pmc $pmc rc=$?" >> $out/summary.txt
done
# condense: per-kernel stats + counter averages for our kernels
python3 - <<PY >> $out/summary.txt
import csv, glob, collections
for f in glob.glob("$out/trace/**/*kernel_stats.csv", recursive=True):
    print("== kernel_stats", f)
    for row in csv.DictReader(open(f)):
        if "srx_" in row.get("Name","") or float(row.get("Percentage",0) or 0) > 2:
            print({k: row[k] for k in row if k in ("Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs")})
for f in glob.glob("$out/pmc_*/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(f)):
        kn = row.get("Kernel_Name","")
        if "srx_" not in kn: continue
        key = (kn.split("(")[0][:60], row["Counter_Name"])
        acc[key][0] += float(row["Counter_Value"]); acc[key][1] += 1
    print("== pmc", f.split("/")[-3] if "/" in f else f)
    for (kn, cn), (s, n) in sorted(acc.items()):
        print(f"{kn:62s} {cn:28s} avg_per_dispatch={s/n:.4g} dispatches={n}")
PY
cat $out/summary.txt
