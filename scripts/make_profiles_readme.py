"""Write profiles/README.md from the evidence files of a round (bench lines, rocprofv3 kernel
statistics, PMC traffic):  python scripts/make_profiles_readme.py r01
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"


def bench(name):
    path = os.path.join(P, f"{tag}_bench_{name}.json")
    if not os.path.exists(path):
        return None
    return json.loads(open(path).read().strip().splitlines()[-1])


rows = [("default", "c4_500x300k banded (the `bench.py` default)"),
        ("default_shuffled", "c4_500x300k banded, point ids relabelled at random (`--shuffle-ids`)"),
        ("default_rows", "c4_500x300k banded, rows of A materialised (`SFM_BA_STATE=0`)"),
        ("c4_500x300k_tracks", "c4_500x300k tracks"), ("c4_500x300k_uniform", "c4_500x300k uniform"),
        ("ladybug50_banded", "ladybug50 banded"), ("ladybug50_tracks", "ladybug50 tracks"), ("ladybug50_uniform", "ladybug50 uniform"),
        ("venice1800_banded", "venice1800 banded"), ("venice1800_tracks", "venice1800 tracks"),
        ("venice1800_uniform", "venice1800 uniform (dense S, 12 600 unknowns)")]
for w in ("c4_500x300k", "venice1800"):
    for v, what in (("loop", "one lap closing a loop"), ("closures", "two laps over the same places")):
        rows.append((f"{w}_{v}", f"{w} {what}, cameras in the folded order"))
        rows.append((f"{w}_{v}_acquisition", f"{w} {what}, cameras in acquisition order (`SFM_BA_REORDER=0`)"))


def rate(g):
    """Stage rate in GB/s of algorithmic bytes; in brackets when the stage forms the rows of A itself
    (fp64-bound: the figure prices replaced work and is not a bandwidth)."""
    if g.get("GBps") is not None:
        return "%.0f" % g["GBps"]
    if g.get("priced_GBps") is not None:
        return "[%.0f]" % g["priced_GBps"]
    return "-"


out = [f"# profiles — round {tag[1:]} evidence (MI355X, one GPU)", "",
       "Produced by `bash scripts/gpu_evidence.sh <tag>` on the GPU box; this file is generated from the files next to it",
       "by `scripts/make_profiles_readme.py`.", "",
       "## Bench lines (`bench.py`, 10 timed Gauss-Newton iterations after 3 warm-up iterations)", "",
       "Stage rates price the bytes the loop has to move (136 / 152 / 120 per observation: B is read out of A and never",
       "written) over the HIP-event time of the stage's kernels.  A stage whose kernels read the rows of A from HBM is",
       "HBM-bound and the figure is its bandwidth.  A stage whose kernels form the rows themselves is bound by the fp64",
       "pipe: its figure is in square brackets, prices the work it replaces and is not a bandwidth (it may exceed 8000).",
       "Where the rows of A are formed inside the assembly kernel, Jacobian and assembly are one stage.", "",
       "| workload | cameras x points x observations | assembly | ms/iteration (with / without event records) | observations/s | reduced solve ms (path) | stage: time us, rate GB/s |",
       "|---|---|---|---|---|---|---|"]
for key, label in rows:
    d = bench(key)
    if d is None:
        continue
    g = d["roofline"]["groups"]
    c = d["config"]
    stages = "; ".join("%s %.0f us, %s" % (nm, 1e3 * gg["ms"], rate(gg)) for nm, gg in g.items())
    order = c.get("camera_order", "")
    out.append("| %s | %d x %d x %d | %s | %.3f / %.3f | %.3g | %.3f (%s) | %s |" % (
        label, c["cameras"], c["points"], c["observations"],
        c.get("assembly", "") + ("; " + order if order.startswith("folded") else ""), d["ms_per_step"], d["ms_per_step_without_events"],
        d["value"], d["solve"]["ms"], d["solve"].get("path", "blocked Cholesky"), stages))
d = bench("default")
if d is not None:
    r = d["roofline"]
    out += ["", "## Default bench line: roofline and CPU baseline", "",
            f"* dominant stage: {r['kernel']}",
            "* bound: %s; achieved %.3g %s = %.1f %% of the %.4g %s peak" % (
                r["bound"], r["achieved"], r["unit"], 100 * r["frac"], r["peak"], r["unit"])]
    if r["bound"] == "mfma":
        out += ["* its algorithmic bytes (the two stages it replaces) retire at %.0f GB/s (%.0f with B priced as a materialised array);"
                " that is a rate of work, not of bytes moved" % (r["achieved_algorithmic"], r["achieved_algorithmic_with_B"])]
    out += ["* measured HBM traffic of that stage (PMC, below): %s" % (
                "%.1f MB per iteration = %.0f GB/s = %.1f %% of the HBM peak" % (
                    r["traffic"] / 1e6, r["groups"][max(r["groups"], key=lambda k: r["groups"][k]["ms"])]["traffic_GBps"],
                    100 * r["groups"][max(r["groups"], key=lambda k: r["groups"][k]["ms"])]["traffic_GBps"] / 8000.0)
                if r.get("traffic") else "not collected for this line")]
    for gname, g in r["groups"].items():
        out.append("* %s (bound: %s): %.1f us, algorithmic %.1f MB (%s GB/s)%s" % (
            gname, g["bound"], 1e3 * g["ms"], g["algorithmic_bytes"] / 1e6, rate(g),
            ", measured traffic %.1f MB (%.0f GB/s)" % (g["traffic"] / 1e6, g["traffic_GBps"]) if g.get("traffic") else ""))
    if r.get("note"):
        out.append("* note: " + r["note"])
    for key, what in (("shuffled_ids", "point ids relabelled at random"), ("tracks_visibility", "track visibility"),
                      ("uniform_visibility", "uniform visibility"), ("loop_visibility", "one lap closing a loop"),
                      ("venice1800_banded", "the 1 M-point workload")):
        u = d.get(key)
        if not u:
            continue
        ug = u["roofline"]["groups"]
        out.append("* sub-record `%s` (%s): %.3f ms per iteration, %.3g observations/s; %s; solve %.3f ms (%s); assembly: %s" % (
            key, what, u["ms_per_step"], u["value"],
            "; ".join("%s %.1f us (%s, %s GB/s)" % (nm, 1e3 * gg["ms"], gg["bound"], rate(gg)) for nm, gg in ug.items()),
            u["solve"]["ms"], u["solve"]["path"], u["assembly"]))
    cb = d.get("cpu_baseline")
    if cb:
        out += ["* CPU baseline (`kind: %s`, %d BLAS threads): %.3g observations/s — %s" % (
            cb["kind"], cb["cores"], cb["value"], cb["sample"])]
    out += ["", "Per-kernel HIP-event times of that line (ms per launch, launches per iteration):", ""]
    for k, v in d["kernels"].items():
        out.append("* `%s`: %.4f ms x %.0f" % (k, v["ms_avg"], v["launches_per_step"]))
summ = os.path.join(P, f"{tag}_c4_banded_kernel_summary.txt")
if os.path.exists(summ):
    out += ["", f"## rocprofv3 `--kernel-trace --stats` of the default bench (`{tag}_c4_banded_kernel_stats.csv`)", "",
            "Average kernel durations agree with the HIP-event times above (the profile covers warm-up, both timed",
            "passes of `bench.py` and the pass that times the small kernels: 33 iterations).", "", "```"]
    out += [line.rstrip()[:150] for line in open(summ)]
    out += ["```"]
reh = bench("rehearsal_2ranks")
if reh is not None:
    out += ["", "## Two-rank rehearsal of the sharded bench on the one GPU (`SFM_BA_BENCH_BACKEND=gloo bench.py --gpus 2 --workload ladybug50 --scaling strong`)", "",
            "* n_gpus %d, scaling %s, final cost %.12g (the one-rank line of the same workload: see `%s_bench_ladybug50_banded.json`);"
            " the ranks share one device and exchange through host copies, so the time (%.3f ms per iteration) means nothing" % (
                reh["n_gpus"], reh["scaling"], reh["final_cost"], tag, reh["ms_per_step"])]
pmc = os.path.join(P, f"{tag}_pmc_traffic.json")
if os.path.exists(pmc):
    doc = json.load(open(pmc))
    out += ["", "## HBM traffic per launch (rocprofv3 `--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`, separate passes)", "",
            "Counters are in KiB. `bytes` = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: on gfx950 FETCH_SIZE tallies a 128-byte",
            "request of a wide streaming read as 64 bytes, so the read side is doubled; for kernels that read with",
            "narrower accesses this is an upper bound (the raw counter is kept).", ""]
    for wl, ks in doc.items():
        out += [f"### {wl}", "", "| kernel | FETCH_SIZE KiB | WRITE_SIZE KiB | corrected MB per launch |", "|---|---|---|---|"]
        for k, v in ks.items():
            out.append("| `%s` | %.4g | %.4g | %.2f |" % (k, v["fetch_kib"], v["write_kib"], v["bytes"] / 1e6))
        out.append("")
tri = os.path.join(P, f"{tag}_triangulate.json")
if os.path.exists(tri):
    t = json.loads(open(tri).read().strip().splitlines()[-1])
    out += ["", "## Two-view triangulation (`bench.py --triangulate 8000000 --steps 20`, inputs resident in HBM)", "",
            "* %d correspondences: %.3g correspondences/s, kernel %.0f us per launch (HIP events)" % (t["n"], t["value"], t["kernel_us"]),
            "* algorithmic 64 B per correspondence -> %.0f GB/s = %.1f %% of the HBM peak: the kernel is bound by the fp64"
            " arithmetic of the Jacobi sweeps (about five sweeps of six column rotations per correspondence), not by memory" % (
                t["roofline"]["achieved"], 100 * t["roofline"]["frac"]),
            "* CPU baseline (`kind: port`): %.3g correspondences/s — %s" % (t["cpu_baseline"]["value"], t["cpu_baseline"]["sample"]),
            "* largest difference from the oracle after rounding to float32: %.1e of the largest coordinate" % t["max_rel_diff_vs_oracle"]]
pnp = os.path.join(P, f"{tag}_pnp.json")
if os.path.exists(pnp):
    t = json.loads(open(pnp).read().strip().splitlines()[-1])
    out += ["", "## RANSAC linear PnP (`bench.py --pnp 20000 --steps 10`, whole call from host arrays)", "",
            "* %d correspondences, 200 minimal samples + inlier counts + refit on %d inliers: %.1f ms per call" % (
                t["n"], t["inliers"], t["ms_per_call"]),
            "* CPU baseline (`kind: port`): %.3g calls/s - %s" % (t["cpu_baseline"]["value"], t["cpu_baseline"]["sample"]),
            "* inliers %d (oracle %d); largest difference of the pose from the oracle: %.1e" % (
                t["inliers"], t["inliers_oracle"], t["max_rel_diff_vs_oracle"])]
for name, title in (("setup", "Handing a problem over as host arrays (`scripts/time_setup.py`, default workload)"),
                    ("dropin", "Drop-in `optimize_using_BA(frames, map_)`, 10 iterations (`scripts/time_dropin.py c4_500x300k`)")):
    path = os.path.join(P, f"{tag}_{name}.txt")
    if os.path.exists(path):
        out += ["", "## " + title, "", "```"] + [line.rstrip() for line in open(path)] + ["```"]
out += ["", "## Files", "",
        f"* `{tag}_bench_*.json` — the JSON line of each bench run",
        f"* `{tag}_c4_banded_kernel_stats.csv`, `{tag}_c4_banded_kernel_summary.txt` — rocprofv3 kernel statistics of the default bench",
        f"* `{tag}_*_pmc_fetch_size.txt`, `{tag}_*_pmc_write_size.txt`, `{tag}_pmc_traffic.json` — PMC passes (default workload, the same with track and with uniform visibility, venice1800 banded)",
        f"* `{tag}_triangulate.json`, `{tag}_pnp.json` — triangulation and PnP benchmark lines",
        f"* `{tag}_setup.txt`, `{tag}_dropin.txt` — host-side timings of plan build / uploads and of the drop-in call",
        "* `README_r01.md`, `README_r02.md` and the `r01_*`, `r02_*` files — the earlier rounds' evidence, for comparison",
        ""]
open(os.path.join(P, "README.md"), "w").write("\n".join(out))
print("wrote profiles/README.md")
