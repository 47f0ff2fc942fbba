#!/bin/bash
# usage: scratch/pmc.sh TAG [bench args...]  -> gpurun_out/pmc_TAG/{mfma,fetch,write}/..., summary gpurun_out/pmc_TAG.csv
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export FPC_STREAMS=1 FPC_NMS_ASIDE=0
for pass in "mfma:SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
  name=${pass%%:*}; ctr=${pass#*:}
  rocprofv3 --pmc $ctr --output-format csv -d $R/gpurun_out/pmc_$TAG/$name -o p -- python3 $R/bench.py --steps 3 --warmup 1 --only-timed --no-timing-events "$@" > /dev/null 2>&1 || echo "pass $name failed"
done
python3 $R/profiles/summarize_pmc.py $(find $R/gpurun_out/pmc_$TAG -name "*counter_collection.csv") > $R/gpurun_out/pmc_$TAG.csv
cut -d, -f1,2,4,5,7,8 $R/gpurun_out/pmc_$TAG.csv | cut -c1-150
