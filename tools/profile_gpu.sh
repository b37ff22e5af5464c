#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): one rocprofv3 --kernel-trace --stats pass and separate --pmc passes of a command.
#   tools/profile_gpu.sh <tag> <command ...>          e.g.  tools/profile_gpu.sh bunny_mk python3 bench.py --steps 3 --warmup 1 --no-cpu
# The program itself follows the tag (never env / bash -c: rocprofv3's preloaded library has initialised the GPU by then).
# Output under gpurun_out/prof_<tag>/; tools/summarize_profile.py turns it into profiles/<name>.json + <name>_kernel_stats.csv.
# Counters are grouped by hardware block budget (MI355X_MICROARCH.md: SQ 8 slots, TCC 4 with FETCH_SIZE = 3 and WRITE_SIZE = 2).
TAG=${1:-cur}
shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf $OUT
mkdir -p $OUT
export TMPDIR=/tmp
# paths in the command are relative to the repository root: make the one that names a file absolute
CMD=()
for a in "$@"; do
  if [ -f "$ROOT/$a" ]; then CMD+=("$ROOT/$a"); else CMD+=("$a"); fi
done
cd /tmp
echo "${CMD[@]}" > $OUT/command.txt
# the tree the counters come from (bench.py only quotes counters whose stamp is the running tree's): recorded HERE, on the box
(cd $ROOT && python3 -c "from goblin_amd import build; print(build.source_stamp())") > $OUT/stamp.txt
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- "${CMD[@]}" > $OUT/stats.log 2>&1 || echo "stats pass failed" >&2
# (the second SQ group: where a wave's cycles go -- parked at s_waitcnt / issue stall / issuing -- and the lanes its VALU instructions had switched on)
for C in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout -k 10 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$N -- "${CMD[@]}" > $OUT/pmc_$N.log 2>&1 || echo "pmc pass $N failed" >&2
done
echo done > $OUT/done
