#!/bin/bash
# usage (on the GPU box): bash profiles/pmc_passes.sh TAG [--frames=N] [bench args...]
#   -> gpurun_out/pmc_TAG/{mfma,fetch,write}/..., summary gpurun_out/pmc_TAG.csv, stderr of each pass in gpurun_out/pmc_TAG/*.err
# Three SEPARATE rocprofv3 --pmc passes (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass), no trace
# domains beside them, kernels alone on the GPU (one stream, NMS in line).
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is the repo copy on the GPU box)}"
TAG=$1; shift
FRAMES=""
if [[ "${1:-}" == --frames=* ]]; then FRAMES=$1; shift; fi
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
export FPC_STREAMS=1 FPC_NMS_ASIDE=0
for pass in "mfma:SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
  name=${pass%%:*}; ctr=${pass#*:}
  rocprofv3 --pmc $ctr --output-format csv -d "$O/$name" -o p -- python3 "$R/bench.py" --steps 3 --warmup 1 --only-timed --no-timing-events "$@" > "$O/$name.out" 2> "$O/$name.err"
done
python3 "$R/profiles/summarize_pmc.py" $FRAMES $(find "$O" -name "*counter_collection.csv") > "$R/gpurun_out/pmc_$TAG.csv"
cut -d, -f1,2,4,5,7,8 "$R/gpurun_out/pmc_$TAG.csv" | cut -c1-150
