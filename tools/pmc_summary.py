import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for r in rows:
    k = r["Kernel_Name"].split("(")[0][-60:]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k, v in sorted(agg.items()):
    print(k, {c: (round(x / cnt[k][c], 1), cnt[k][c]) for c, x in v.items()})
