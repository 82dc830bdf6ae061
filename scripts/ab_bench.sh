#!/bin/bash
# usage: scripts/ab_bench.sh <tag> "<env assignments A>" "<env assignments B>" ... [-- bench args]
# Runs the default bench line once per environment (on the GPU box) and prints ms per iteration and the
# slowest kernels of each: quick A/B comparisons of diagnostic builds / switches.
TAG=$1; shift
ARGS="--steps 10 --warmup 3 --no-cpu-baseline --no-extra-records"
i=0
for E in "$@"; do
  i=$((i+1))
  env $E timeout -k 10 200 python3 bench.py $ARGS > gpurun_out/${TAG}_$i.json 2> gpurun_out/${TAG}_$i.err || { echo "run $i ($E) failed"; tail -3 gpurun_out/${TAG}_$i.err; exit 1; }
  python3 - "$E" gpurun_out/${TAG}_$i.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
k = sorted(d["kernels"].items(), key=lambda kv: -kv[1]["ms_avg"] * kv[1]["launches_per_step"])[:5]
print("[%s] %s: %.4f ms (%.4f without events) | " % (sys.argv[1], d["config"]["workload"].split(":")[0], d["ms_per_step"], d["ms_per_step_without_events"])
      + ", ".join("%s %.1f" % (n, v["ms_avg"] * v["launches_per_step"] * 1e3) for n, v in k))
PY
done
