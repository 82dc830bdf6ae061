#!/bin/bash
# usage: scripts/gpu_evidence.sh <tag> [round tag, default r01]   (on the GPU box, from the repo root)
# Default bench line, the rocprofv3 kernel statistics of the same command, the two PMC traffic
# passes, and one bench line per other workload.  Everything lands under gpurun_out/<tag>_*.
set -o pipefail
TAG=$1
RT=${2:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd $R
bash scripts/gpu_prof.sh ${TAG} > gpurun_out/${TAG}_prof_summary.txt || exit 2
bash scripts/gpu_pmc.sh ${TAG}_fetch FETCH_SIZE $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_pmc_fetch.txt || exit 3
bash scripts/gpu_pmc.sh ${TAG}_write WRITE_SIZE $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_pmc_write.txt || exit 4
# the default bench line quotes the traffic of these two passes: hand them over before it runs
python3 scripts/pmc_to_json.py gpurun_out/${TAG}_pmc_fetch.txt gpurun_out/${TAG}_pmc_write.txt c4_500x300k/banded profiles/${RT}_pmc_traffic.json > /dev/null || exit 4
timeout -k 10 300 python3 bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err || exit 1
for W in "ladybug50 banded" "ladybug50 uniform" "c4_500x300k uniform" "venice1800 banded"; do
  set -- $W
  timeout -k 10 400 python3 bench.py --workload $1 --visibility $2 --no-cpu-baseline > gpurun_out/${TAG}_bench_$1_$2.json 2> gpurun_out/${TAG}_bench_$1_$2.err || exit 5
done
timeout -k 10 300 python3 bench.py --triangulate 8000000 --steps 20 > gpurun_out/${TAG}_triangulate.json 2> gpurun_out/${TAG}_triangulate.err || exit 6
timeout -k 10 300 python3 bench.py --pnp 20000 --steps 10 > gpurun_out/${TAG}_pnp.json 2> gpurun_out/${TAG}_pnp.err || exit 7
timeout -k 10 300 python3 scripts/time_setup.py > gpurun_out/${TAG}_setup.txt 2>&1 || exit 8
timeout -k 10 300 python3 scripts/time_dropin.py c4_500x300k > gpurun_out/${TAG}_dropin.txt 2>&1 || exit 9
echo evidence done
