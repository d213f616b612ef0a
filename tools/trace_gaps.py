"""Per-kernel duration and the gap to the previous kernel on the stream, from a rocprofv3 --kernel-trace CSV.
usage: python tools/trace_gaps.py <kernel_trace.csv> [min_calls]   (kernels of the steady state: those with >= min_calls dispatches)"""
import csv, re, sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
min_calls = int(sys.argv[2]) if len(sys.argv) > 2 else 200
def short(n):
    n = re.sub(r"\(anonymous namespace\)::|jh::|void ", "", n)
    return n.split("(")[0][:60]
dur, gap, cnt = defaultdict(float), defaultdict(float), defaultdict(int)
prev_end = None
for s, e, n in rows:
    k = short(n)
    dur[k] += e - s
    if prev_end is not None and s - prev_end < 200_000:   # gaps beyond 0.2 ms are host pauses, not the loop
        gap[k] += s - prev_end
    cnt[k] += 1
    prev_end = max(prev_end or 0, e)
tot = sum(dur.values())
print(f"{'kernel':60s} {'calls':>7s} {'avg us':>8s} {'gap before us':>13s} {'share':>6s}")
for k in sorted(dur, key=lambda k: -dur[k]):
    if cnt[k] >= min_calls:
        print(f"{k:60s} {cnt[k]:7d} {dur[k] / cnt[k] / 1e3:8.2f} {gap[k] / cnt[k] / 1e3:13.2f} {dur[k] / tot:6.3f}")
