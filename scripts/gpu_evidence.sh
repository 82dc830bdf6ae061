#!/bin/bash
# usage: scripts/gpu_evidence.sh <tag> [round tag, default r03]   (on the GPU box, from the repo root)
# Default bench line, the rocprofv3 kernel statistics of the same command, PMC traffic passes
# (default workload, the same with track and with uniform visibility, venice1800 banded), one bench
# line per other workload, a two-rank rehearsal of the sharded bench.  Everything lands under
# gpurun_out/<tag>_*.
set -o pipefail
TAG=$1
RT=${2:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd $R
bash scripts/gpu_prof.sh ${TAG} --no-extra-records > gpurun_out/${TAG}_prof_summary.txt || exit 2
# the same for uniform visibility (camera-pair assembly from the state, dense factorisation)
bash scripts/gpu_prof.sh ${TAG}u --visibility uniform --no-extra-records --no-cpu-baseline > gpurun_out/${TAG}_uniform_prof_summary.txt || exit 2
pmc_pair() {  # <name> <traffic key> <bench args...>
  local NAME=$1 KEY=$2; shift; shift
  bash scripts/gpu_pmc.sh ${TAG}_${NAME}_fetch FETCH_SIZE $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-records "$@" > gpurun_out/${TAG}_${NAME}_pmc_fetch.txt || return 3
  bash scripts/gpu_pmc.sh ${TAG}_${NAME}_write WRITE_SIZE $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-records "$@" > gpurun_out/${TAG}_${NAME}_pmc_write.txt || return 4
  python3 scripts/pmc_to_json.py gpurun_out/${TAG}_${NAME}_pmc_fetch.txt gpurun_out/${TAG}_${NAME}_pmc_write.txt $KEY profiles/${RT}_pmc_traffic.json > /dev/null || return 4
}
# the bench lines quote the traffic of these passes: hand them over before they run
pmc_pair c4_banded c4_500x300k/banded || exit 3
pmc_pair c4_tracks c4_500x300k/tracks --visibility tracks || exit 3
pmc_pair c4_uniform c4_500x300k/uniform --visibility uniform || exit 3
pmc_pair venice_banded venice1800/banded --workload venice1800 || exit 3
timeout -k 10 400 python3 bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err || exit 1
# the same line with the rows of A written by the Jacobian kernel and read back twice (three separate
# HBM-bound stages, priced separately)
SFM_BA_STATE=0 timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-extra-records > gpurun_out/${TAG}_bench_default_rows.json 2> gpurun_out/${TAG}_bench_default_rows.err || exit 1
timeout -k 10 300 python3 bench.py --shuffle-ids --no-cpu-baseline --no-extra-records > gpurun_out/${TAG}_bench_default_shuffled.json 2> gpurun_out/${TAG}_bench_default_shuffled.err || exit 1
for W in "ladybug50 banded" "ladybug50 uniform" "ladybug50 tracks" "c4_500x300k uniform" "c4_500x300k tracks" "venice1800 banded" "venice1800 tracks" "venice1800 uniform"; do
  set -- $W
  timeout -k 10 500 python3 bench.py --workload $1 --visibility $2 --no-cpu-baseline --no-extra-records > gpurun_out/${TAG}_bench_$1_$2.json 2> gpurun_out/${TAG}_bench_$1_$2.err || exit 5
done
# sequences that revisit places (one lap closing a loop; two laps): cameras in the folded order
# (near points fused, far points camera-pair, inside one plan), and the same in acquisition order
for W in "c4_500x300k loop" "venice1800 loop" "c4_500x300k closures" "venice1800 closures"; do
  set -- $W
  timeout -k 10 500 python3 bench.py --workload $1 --visibility $2 --no-cpu-baseline --no-extra-records > gpurun_out/${TAG}_bench_$1_$2.json 2> gpurun_out/${TAG}_bench_$1_$2.err || exit 5
  SFM_BA_REORDER=0 timeout -k 10 500 python3 bench.py --workload $1 --visibility $2 --no-cpu-baseline --no-extra-records > gpurun_out/${TAG}_bench_$1_$2_acquisition.json 2> gpurun_out/${TAG}_bench_$1_$2_acquisition.err || exit 5
done
# two ranks on this one GPU over gloo: exercises the sharded bench path (the times mean nothing)
SFM_BA_BENCH_BACKEND=gloo timeout -k 10 300 python3 bench.py --gpus 2 --workload ladybug50 --scaling strong --no-cpu-baseline > gpurun_out/${TAG}_bench_rehearsal_2ranks.json 2> gpurun_out/${TAG}_bench_rehearsal_2ranks.err || exit 10
timeout -k 10 300 python3 bench.py --triangulate 8000000 --steps 20 > gpurun_out/${TAG}_triangulate.json 2> gpurun_out/${TAG}_triangulate.err || exit 6
timeout -k 10 300 python3 bench.py --pnp 20000 --steps 10 > gpurun_out/${TAG}_pnp.json 2> gpurun_out/${TAG}_pnp.err || exit 7
timeout -k 10 300 python3 scripts/time_setup.py > gpurun_out/${TAG}_setup.txt 2>&1 || exit 8
timeout -k 10 300 python3 scripts/time_dropin.py c4_500x300k > gpurun_out/${TAG}_dropin.txt 2>&1 || exit 9
echo evidence done
