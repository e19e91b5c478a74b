"""Dev helper: per-kernel averages of rocprofv3 --pmc counter_collection.csv files under a directory, for kernels whose name
contains a substring; the largest-grid dispatch of each kernel is listed separately (the main round of a multi-launch op).
python tools/pmc_summary.py <dir> [substring]"""
import csv, glob, collections, re, sys

root = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else "srx_"
for f in sorted(glob.glob(root + "/pmc_*/**/*counter_collection.csv", recursive=True)):
    per = collections.defaultdict(lambda: collections.defaultdict(list))  # kernel -> counter -> [(grid, value)]
    for row in csv.DictReader(open(f)):
        kn = row.get("Kernel_Name", "")
        if sub not in kn:
            continue
        m = re.search(r"(srx_\w+(<\d+>)?)", kn)
        per[m.group(1) if m else kn[:60]][row["Counter_Name"]].append((int(row.get("Grid_Size", 0) or 0), float(row["Counter_Value"])))
    print("==", f.split("/")[-3])
    for kn in sorted(per):
        for cn in sorted(per[kn]):
            v = per[kn][cn]
            gmax = max(g for g, _ in v)
            big = [x for g, x in v if g == gmax]
            print(f"{kn:36s} {cn:28s} avg={sum(x for _, x in v) / len(v):.4g} n={len(v)}  largest-grid({gmax}) avg={sum(big) / len(big):.4g} n={len(big)}")
