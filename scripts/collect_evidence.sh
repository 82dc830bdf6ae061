#!/bin/bash
# usage: scripts/collect_evidence.sh <run tag> <round tag>   e.g. r01ev r01
# Copies what scripts/gpu_evidence.sh left under gpurun_out/ into profiles/ and regenerates profiles/README.md.
set -e
RUN=$1; RT=$2
cd "$(dirname "$0")/.."
cp "$(ls -t gpurun_out/prof_${RUN}/*/*kernel_stats.csv | head -1)" profiles/${RT}_c4_banded_kernel_stats.csv
cp gpurun_out/${RUN}_prof_summary.txt profiles/${RT}_c4_banded_kernel_summary.txt
[ -f gpurun_out/${RUN}_uniform_prof_summary.txt ] && cp gpurun_out/${RUN}_uniform_prof_summary.txt profiles/${RT}_c4_uniform_kernel_summary.txt
for n in c4_banded c4_tracks c4_uniform venice_banded; do
  cp gpurun_out/${RUN}_${n}_pmc_fetch.txt profiles/${RT}_${n}_pmc_fetch_size.txt
  cp gpurun_out/${RUN}_${n}_pmc_write.txt profiles/${RT}_${n}_pmc_write_size.txt
done
for f in gpurun_out/${RUN}_bench_*.json; do b=$(basename $f); cp $f profiles/${b/${RUN}_/${RT}_}; done
for x in triangulate.json pnp.json; do [ -f gpurun_out/${RUN}_$x ] && cp gpurun_out/${RUN}_$x profiles/${RT}_$x; done
[ -f gpurun_out/${RUN}_setup.txt ] && grep "^c4_" gpurun_out/${RUN}_setup.txt > profiles/${RT}_setup.txt
[ -f gpurun_out/${RUN}_dropin.txt ] && grep " rep [0-9]*: optimize_using_BA" gpurun_out/${RUN}_dropin.txt > profiles/${RT}_dropin.txt
python3 scripts/pmc_to_json.py profiles/${RT}_c4_banded_pmc_fetch_size.txt profiles/${RT}_c4_banded_pmc_write_size.txt c4_500x300k/banded profiles/${RT}_pmc_traffic.json
python3 scripts/pmc_to_json.py profiles/${RT}_c4_tracks_pmc_fetch_size.txt profiles/${RT}_c4_tracks_pmc_write_size.txt c4_500x300k/tracks profiles/${RT}_pmc_traffic.json
python3 scripts/pmc_to_json.py profiles/${RT}_c4_uniform_pmc_fetch_size.txt profiles/${RT}_c4_uniform_pmc_write_size.txt c4_500x300k/uniform profiles/${RT}_pmc_traffic.json
python3 scripts/pmc_to_json.py profiles/${RT}_venice_banded_pmc_fetch_size.txt profiles/${RT}_venice_banded_pmc_write_size.txt venice1800/banded profiles/${RT}_pmc_traffic.json
python3 scripts/make_profiles_readme.py ${RT}
