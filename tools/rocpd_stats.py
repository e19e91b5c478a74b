"""Dev helper: per-kernel summary (calls, average / min / max / total duration) of a rocprofv3 rocpd database
(`rocprofv3 --kernel-trace -d DIR -o NAME` writes DIR/NAME_results.db).  python tools/rocpd_stats.py <db> [top]"""
import sqlite3, sys

con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
name = "kernel_name" if "kernel_name" in cols else [c for c in cols if "name" in c][0]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 14
q = f"""select s.{name}, count(*), avg(k.end - k.start), min(k.end - k.start), max(k.end - k.start), sum(k.end - k.start)
        from rocpd_kernel_dispatch k join rocpd_info_kernel_symbol s on k.kernel_id = s.id group by s.{name} order by 6 desc limit {top}"""
print(f"{'kernel':90s} {'calls':>6s} {'avg us':>10s} {'min us':>10s} {'max us':>10s} {'total ms':>10s}")
for r in cur.execute(q):
    print(f"{r[0][:90]:90s} {r[1]:6d} {r[2] / 1e3:10.1f} {r[3] / 1e3:10.1f} {r[4] / 1e3:10.1f} {r[5] / 1e6:10.2f}")
