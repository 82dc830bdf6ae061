"""Durations of the dense factorisation's launches, in order, from a rocprofv3 kernel trace directory:
python scripts/trace_chol.py <dir> (last solve of the run)"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "k_chol" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_chol_diag" in r["Kernel_Name"]][-1]
last = rows[idx:]
out = []
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    out.append((r["Kernel_Name"].split("(")[0].replace("sfmba::", ""), int(r["Grid_Size_X"]) // 256, (e - s) / 1e3, s, e))
steps = [o for o in out if o[0] == "k_chol_step"]
print("launches", len(out), "total %.1f us" % ((out[-1][4] - out[0][3]) / 1e3), " steps: %d, sum %.1f us" % (len(steps), sum(o[2] for o in steps)))
print("step durations (us), grid:", " ".join("%.1f/%d" % (o[2], o[1]) for o in steps))
back = [o for o in out if o[0].startswith("k_chol_back")]
print("back: %d launches, sum %.1f us" % (len(back), sum(o[2] for o in back)))
