"""Timeline of ONE linear solve from a rocprofv3 --kernel-trace CSV: every kernel between the n-th and (n+1)-th jagged_copy_kernel
launch (the start of a BiCGStab solve) with its start offset, duration and the gap to the previous kernel's end.
usage: python tools/trace_timeline.py <kernel_trace.csv> [solve index = 10] [max rows = 80]"""
import csv, re, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        if r.get("Start_Timestamp"):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
k = int(sys.argv[2]) if len(sys.argv) > 2 else 10
mx = int(sys.argv[3]) if len(sys.argv) > 3 else 80
starts = [i for i, r in enumerate(rows) if "jagged_copy_kernel" in r[2]]
i0, i1 = starts[k], starts[k + 1]
# include the kernels of the Newton step in front of the solve (assembly, factor)
j = i0
while j > 0 and rows[j][0] - rows[j - 1][1] < 300_000 and i0 - j < 12:
    j -= 1
t0 = rows[j][0]
prev = None
def short(n):
    n = re.sub(r"\(anonymous namespace\)::|jh::|void ", "", n)
    return n.split("(")[0][:58]
print(f"{'start us':>9s} {'dur us':>8s} {'gap us':>8s}  kernel")
for n, (s, e, name) in enumerate(rows[j:i1]):
    if n >= mx and n < (i1 - j) - 25:
        if n == mx: print("   ...")
        prev = e
        continue
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.2f} {((s - prev) / 1e3 if prev else 0):8.2f}  {short(name)}")
    prev = e
print(f"solve span {(rows[i1 - 1][1] - rows[i0][0]) / 1e3:.1f} us, {i1 - i0} kernels")
