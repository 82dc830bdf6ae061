"""Merge the per-kernel FETCH_SIZE / WRITE_SIZE averages printed by scripts/pmc_summary.py (two
rocprofv3 --pmc passes) into one JSON file of HBM traffic per launch.

usage: python scripts/pmc_to_json.py <fetch.txt> <write.txt> <workload key> <out.json>

Both counters are in KiB.  On gfx950 FETCH_SIZE tallies a 128-byte request of a wide (16 bytes per
lane) streaming read as 64 bytes, so the read side is doubled: `bytes` = (2 FETCH_SIZE +
WRITE_SIZE) * 1024 per launch.  Kernels that read with narrower accesses are over-counted by this
correction by at most the raw read volume, which is kept in the file as `fetch_kib`.
"""
import json, re, sys

def parse(path, name):
    out = {}
    for line in open(path):
        m = re.search(r"sfmba::(\w+)(?:<[^>]*>)?\s+%s=([0-9.e+-]+)\(n=(\d+)\)" % name, line.strip())
        if m:
            out[m.group(1)] = (float(m.group(2)), int(m.group(3)))
    return out

fetch, write = parse(sys.argv[1], "FETCH_SIZE"), parse(sys.argv[2], "WRITE_SIZE")
key, dst = sys.argv[3], sys.argv[4]
try:
    doc = json.load(open(dst))
except FileNotFoundError:
    doc = {}
doc[key] = {
    k: {"fetch_kib": fetch[k][0], "write_kib": write.get(k, (0.0, 0))[0], "launches_sampled": fetch[k][1],
        "bytes": (2.0 * fetch[k][0] + write.get(k, (0.0, 0))[0]) * 1024.0}
    for k in sorted(fetch)
}
json.dump(doc, open(dst, "w"), indent=1, sort_keys=True)
print("wrote", dst, "with", len(doc[key]), "kernels for", key)
