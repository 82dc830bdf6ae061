"""Average PMC counter value per dispatch for each kernel (rocprofv3 --pmc csv output)."""
import csv, glob, sys, collections
d = sys.argv[1]
files = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
flt = sys.argv[2] if len(sys.argv) > 2 else "sfmba"
for k in sorted(acc):
    if flt not in k:
        continue
    print(k, " ".join("%s=%.4g(n=%d)" % (c, sum(v) / len(v), len(v)) for c, v in sorted(acc[k].items())))
