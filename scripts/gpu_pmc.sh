#!/bin/bash
# usage: scripts/gpu_pmc.sh <tag> "<counter list>" <python script + args>   one PMC pass, per-kernel averages
set -o pipefail
TAG=$1; CNT=$2; shift; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp && timeout -s KILL 300 rocprofv3 --pmc $CNT --output-format csv -d $R/gpurun_out/pmc_$TAG -- python3 "$@" > $R/gpurun_out/pmc_$TAG.log 2>&1
rc=$?
cd $R && python3 scripts/pmc_summary.py gpurun_out/pmc_$TAG sfmba
exit $rc
