"""Condense gpurun_out/prof_<tag>/ (written by tools/gpu_profile.sh) into profiles/<name>_rocprofv3_summary.txt,
profiles/<name>_kernel_stats.csv and profiles/traffic.json (HBM bytes per launch of the dominant kernel, used by
bench.py's roofline.traffic).  FETCH_SIZE is doubled: on gfx950 it reports exactly half the bytes of a wide coalesced
streaming read (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact.  Both are in KiB."""
import collections, csv, glob, json, shutil, sys

tag, name = sys.argv[1], sys.argv[2]
out = f"gpurun_out/prof_{tag}"


def short(kn):
    for n in ("srx_wave_kernel", "srx_score_kernel", "srx_merge_wave_kernel", "srx_merge_kernel"):
        if n in kn:
            return n
    return None


lines = []
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    shutil.copy(f, f"profiles/{name}_kernel_stats.csv")
    for row in csv.DictReader(open(f)):
        n = short(row["Name"])
        if n:
            lines.append(f"kernel_stats {n:18s} calls={row['Calls']} avg_ns={float(row['AverageNs']):.0f} "
                         f"min_ns={row['MinNs']} max_ns={row['MaxNs']} pct={row['Percentage']}")
# per-launch durations of the dominant kernel in dispatch order: 3 warm-up, 20 timed, 3 PCIe-inclusive launches
for f in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(f)) if "srx_wave_kernel" in r["Kernel_Name"]]
    if len(d) >= 23:
        timed = d[3:23]
        lines.append(f"kernel_trace srx_wave_kernel per launch (ms), dispatch order: " + " ".join(f"{x:.3f}" for x in d))
        lines.append(f"kernel_trace srx_wave_kernel timed launches 4..23: avg_ms={sum(timed) / len(timed):.4f} min_ms={min(timed):.4f} max_ms={max(timed):.4f}")
pm = collections.defaultdict(dict)
for f in sorted(glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(f)):
        kn = short(row.get("Kernel_Name", ""))
        if kn:
            key = (kn, row["Counter_Name"])
            acc[key][0] += float(row["Counter_Value"])
            acc[key][1] += 1
    for (kn, cn), (s, n) in sorted(acc.items()):
        pm[kn][cn] = s / n
        lines.append(f"pmc {kn:18s} {cn:26s} avg_per_dispatch={s / n:.6g} dispatches={n}")
w = pm["srx_wave_kernel"]
hbm = (2 * w.get("FETCH_SIZE", 0) + w.get("WRITE_SIZE", 0)) * 1024
lines.append(f"derived srx_wave_kernel hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 = {hbm:.6g}")
open(f"profiles/{name}_rocprofv3_summary.txt", "w").write(
    "# rocprofv3 --kernel-trace --stats of: python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline (26 launches: 3 warm-up + "
    "20 timed + 3 PCIe-inclusive), then separate --pmc passes of: python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline "
    "(9 launches each); workload c3, 1 x MI355X\n"
    + "\n".join(lines) + "\n")
json.dump({"c3@1": {"hbm_bytes_per_launch": hbm, "fetch_size_kib": w.get("FETCH_SIZE", 0),
                    "write_size_kib": w.get("WRITE_SIZE", 0),
                    "note": "srx_wave_kernel, average per launch; FETCH_SIZE doubled per the gfx950 correction"}},
          open("profiles/traffic.json", "w"), indent=1)
print("\n".join(lines[:3]))
print(lines[-1])
