"""Print the top rows of a rocprofv3 kernel_stats.csv (first one found under the given directory)."""
import csv, glob, sys
d = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 16
f = sorted(glob.glob(d + "/**/*kernel_stats.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:top]:
    print("%-58s calls %6s total_ms %9.3f avg_us %9.2f pct %6.2f" % (r["Name"][:58], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
