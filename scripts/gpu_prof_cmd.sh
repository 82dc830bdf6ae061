#!/bin/bash
# usage: scripts/gpu_prof_cmd.sh <tag> <top> <python script and args...>  (rocprofv3 kernel stats of any python command)
set -o pipefail
TAG=$1; TOP=$2; shift; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 "$@" > $R/gpurun_out/prof_$TAG.log 2>&1
rc=$?
cd $R && grep -E "^n=|^\{" gpurun_out/prof_$TAG.log | cut -c1-300; python3 scripts/prof_summary.py gpurun_out/prof_$TAG $TOP
exit $rc
