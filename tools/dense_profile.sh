#!/bin/bash
# Dev helper (GPU box): rocprofv3 kernel trace + PMC passes of tools/bench_dense.py; summary -> gpurun_out/prof_<tag>/summary.txt
# usage: bash tools/dense_profile.sh <tag> <n_docs> <dim> <nq> <k>
cd ${GRAFT_REPO_ROOT:-.}
tag=$1; shift
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
echo "$@" > $out/args.txt
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 tools/bench_dense.py $@ > $out/trace.log 2>&1
echo "trace rc=$?" > $out/summary.txt
for pmc in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_I8 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $pmc | tr ' ' '_' | cut -c1-40)
  timeout -k 10 300 rocprofv3 --pmc $pmc --output-format csv -d $out/pmc_$name -- python3 tools/bench_dense.py $@ > $out/pmc_$name.log 2>&1
  echo "pmc $pmc rc=$?" >> $out/summary.txt
done
python3 - <<PY >> $out/summary.txt
import csv, glob, collections
for f in glob.glob("$out/trace/**/*kernel_stats.csv", recursive=True):
    print("== kernel_stats", f)
    for row in csv.DictReader(open(f)):
        if "srx_" in row.get("Name","") or float(row.get("Percentage",0) or 0) > 2:
            print({k: row[k] for k in row if k in ("Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs")})
for f in sorted(glob.glob("$out/pmc_*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: [0.0, 0, 0.0])
    for row in csv.DictReader(open(f)):
        kn = row.get("Kernel_Name","")
        if "srx_dense" not in kn: continue
        key = (kn.split("(")[0][:70], row["Counter_Name"])
        v = float(row["Counter_Value"])
        acc[key][0] += v; acc[key][1] += 1; acc[key][2] = max(acc[key][2], v)
    print("== pmc", f.split("/")[-3] if "/" in f else f)
    for (kn, cn), (s, n, mx) in sorted(acc.items()):
        print(f"{kn:72s} {cn:28s} avg_per_dispatch={s/n:.4g} max={mx:.4g} dispatches={n}")
PY
find $out -name '*.csv' -size +2M -delete
cat $out/summary.txt
