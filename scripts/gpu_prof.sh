#!/bin/bash
# usage: scripts/gpu_prof.sh <tag> [bench args...]   (run on the GPU box from the repo root)
set -o pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py "$@" > $R/gpurun_out/prof_$TAG.log 2>&1
rc=$?
cd $R && python3 scripts/prof_summary.py gpurun_out/prof_$TAG 14
grep '^{' gpurun_out/prof_$TAG.log | tail -1 > gpurun_out/prof_$TAG.json
exit $rc
