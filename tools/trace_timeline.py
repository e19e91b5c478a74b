"""Dev helper: timeline of a rocprofv3 --kernel-trace CSV (which kernel ran when, on which queue): per step the busy time
of every queue, the overlap between queues and the idle gaps.  usage: python tools/trace_timeline.py <kernel_trace.csv> [n_last_kernels]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 120
key_s = next(k for k in rows[0] if "start" in k.lower())
key_e = next(k for k in rows[0] if "end" in k.lower())
key_q = next((k for k in rows[0] if "queue" in k.lower()), None)
key_n = next(k for k in rows[0] if "kernel_name" in k.lower() or k.lower() == "name")
rows.sort(key=lambda r: int(r[key_s]))
tail = rows[-n_last:]
t0 = int(tail[0][key_s])
print(f"{'start_us':>10} {'dur_us':>8} {'gap_us':>7} queue  kernel")
prev_end = defaultdict(lambda: None)
all_prev_end = None
for r in tail:
    s, e = int(r[key_s]), int(r[key_e])
    q = r[key_q] if key_q else "0"
    gap = "" if all_prev_end is None else f"{(s - all_prev_end) / 1e3:7.1f}"
    print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f} {gap:>7} {q:>5}  {r[key_n][:70]}")
    all_prev_end = e if all_prev_end is None else max(all_prev_end, e)
# summary by kernel over the tail
acc = defaultdict(lambda: [0, 0.0])
for r in tail:
    a = acc[(r[key_q] if key_q else "0", r[key_n][:60])]
    a[0] += 1
    a[1] += (int(r[key_e]) - int(r[key_s])) / 1e3
print("--- totals over the window")
for (q, n), (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"queue {q:>4} {c:4d} x {t / c:8.1f} us  {n}")
span = (max(int(r[key_e]) for r in tail) - t0) / 1e3
print(f"window {span:.1f} us")
