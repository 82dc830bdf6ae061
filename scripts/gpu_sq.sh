#!/bin/bash
# usage: scripts/gpu_sq.sh <tag> [bench args...]   SQ counter passes of the bench (one rocprofv3 run per pass)
set -o pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
bash $R/scripts/gpu_pmc.sh ${TAG}_sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" || exit 1
bash $R/scripts/gpu_pmc.sh ${TAG}_sq2 "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" || exit 2
bash $R/scripts/gpu_pmc.sh ${TAG}_grbm "GRBM_GUI_ACTIVE GRBM_COUNT" $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" || exit 3
