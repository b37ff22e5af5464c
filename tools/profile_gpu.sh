#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): kernel-trace stats pass + separate PMC passes of the headline bench command.
#   tools/profile_gpu.sh <tag> [schedule]
# Output under gpurun_out/prof_<tag>/; tools/summarize_profile.py turns it into profiles/*.json + *.csv.
set -e
TAG=${1:-cur}
SCHED=${2:-auto}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD="python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu --schedule $SCHED"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/stats.log 2>&1
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE"; do
  N=$(echo $C | tr ' ' '_')
  timeout -k 10 240 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$N -- $CMD > $OUT/pmc_$N.log 2>&1
done
echo done > $OUT/done
