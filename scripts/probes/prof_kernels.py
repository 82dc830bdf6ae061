"""Average duration per kernel (and grid size) out of a rocprofv3 results database:
python scripts/probes/prof_kernels.py <results.db> [name filter]"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = db.execute("select name, grid_x, workgroup_x, duration from kernels where name like ? order by start",
                  (f"%{flt}%",))
agg = collections.defaultdict(list)
for name, g, w, d in rows:
    short = name.split("(")[0].split("::")[-1]
    agg[(short, g // w)].append(d)
for k, v in sorted(agg.items()):
    print(f"{k[0]:28s} wgs {k[1]:6d}  launches {len(v):5d}  avg {sum(v) / len(v) / 1e3:9.1f} us")
