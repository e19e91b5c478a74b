"""Condense gpurun_out/prof_<tag>/ (written by tools/gpu_profile.sh) into profiles/<name>_rocprofv3_summary.txt,
profiles/<name>_kernel_stats.csv and an entry of profiles/traffic.json (HBM bytes per launch of the dominant kernel,
used by bench.py's roofline.traffic together with the sha256 of the kernel source that was profiled).

    python tools/summarize_profile.py <tag> <name> [<workload key, default c3@1>] [<dominant kernel, default srx_wave_kernel>]

FETCH_SIZE is doubled: on gfx950 it reports exactly half the bytes of a wide coalesced streaming read
(MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact.  Both are in KiB."""
import collections, csv, glob, hashlib, json, os, shutil, sys

tag, name = sys.argv[1], sys.argv[2]
wkey = sys.argv[3] if len(sys.argv) > 3 else "c3@1"
dom = sys.argv[4] if len(sys.argv) > 4 else "srx_wave_kernel"
out = f"gpurun_out/prof_{tag}"
sys.path.insert(0, ".")
import sparse_rx  # noqa: E402  (only for the source hash; no GPU needed)


def short(kn):
    for n in ("srx_wave_kernel", "srx_score_kernel", "srx_merge_wave_kernel", "srx_merge_kernel"):
        if n in kn:
            return n
    return None


def newest_per_dir(pattern):
    """gpurun merges every call's files into the same local directory: keep the newest file of each PMC pass."""
    best = {}
    for f in glob.glob(pattern, recursive=True):
        d = f.split("/")[2]
        if d not in best or os.path.getmtime(f) > os.path.getmtime(best[d]):
            best[d] = f
    return sorted(best.values())


lines = []
for f in newest_per_dir(out + "/trace/**/*kernel_stats.csv"):
    shutil.copy(f, f"profiles/{name}_kernel_stats.csv")
    for row in csv.DictReader(open(f)):
        n = short(row["Name"])
        if n:
            lines.append(f"kernel_stats {n:18s} calls={row['Calls']} avg_ns={float(row['AverageNs']):.0f} "
                         f"min_ns={row['MinNs']} max_ns={row['MaxNs']} pct={row['Percentage']}")
# per-launch durations of the dominant kernel in dispatch order: 3 warm-up, 20 timed, then the PCIe-inclusive launches
for f in newest_per_dir(out + "/trace/**/*kernel_trace.csv"):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(f)) if dom in r["Kernel_Name"]]
    if len(d) >= 23:
        timed = d[3:23]
        lines.append(f"kernel_trace {dom} per launch (ms), dispatch order: " + " ".join(f"{x:.3f}" for x in d))
        lines.append(f"kernel_trace {dom} timed launches 4..23: avg_ms={sum(timed) / len(timed):.4f} min_ms={min(timed):.4f} max_ms={max(timed):.4f}")
pm = collections.defaultdict(dict)
for f in newest_per_dir(out + "/pmc_*/**/*counter_collection.csv"):
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
w = pm[dom]
hbm = (2 * w.get("FETCH_SIZE", 0) + w.get("WRITE_SIZE", 0)) * 1024
lines.append(f"derived {dom} hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 = {hbm:.6g}")
sha = sparse_rx._capi.kernel_sources_sha256()
lines.append(f"kernel sources sha256 (csrc/*.hip + srx_common.h) = {sha}")
args = open(out + "/args.txt").read().strip() if os.path.exists(out + "/args.txt") else ""
open(f"profiles/{name}_rocprofv3_summary.txt", "w").write(
    f"# rocprofv3 --kernel-trace --stats of: python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline {args} (3 warm-up + 20 timed "
    "launches, then the PCIe-inclusive loop), then separate --pmc passes of: python3 bench.py --steps 5 --warmup 1 "
    f"--no-cpu-baseline {args}; workload key {wkey}, 1 x MI355X\n"
    + "\n".join(lines) + "\n")
tj = {}
if os.path.exists("profiles/traffic.json"):
    tj = json.load(open("profiles/traffic.json"))
tj[wkey] = {"hbm_bytes_per_launch": hbm, "fetch_size_kib": w.get("FETCH_SIZE", 0), "write_size_kib": w.get("WRITE_SIZE", 0),
            "kernel": dom, "kernel_src_sha256": sha,
            "note": f"{dom}, average per launch over the PMC pass; FETCH_SIZE doubled per the gfx950 correction"}
json.dump(tj, open("profiles/traffic.json", "w"), indent=1)
print("\n".join(lines[:3]))
print(lines[-2])
