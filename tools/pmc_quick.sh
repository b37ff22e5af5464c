#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): a few PMC passes of the headline workload under one schedule, summarised per kernel.
#   tools/pmc_quick.sh <tag> <schedule> [extra bench.py args]
set -e
TAG=${1:-cur}
SCHED=${2:-auto}
shift 2 || true
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmcq_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD="python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu --schedule $SCHED $@"
# counter groups, one pass each: PMC_GROUPS="A B;C D" overrides the default set
DEFAULT_GROUPS="TCC_HIT_sum TCC_MISS_sum;SQ_INSTS_VALU SQ_ACTIVE_INST_VALU;GRBM_GUI_ACTIVE SQ_WAVES;SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR;SQ_INSTS_SALU SQ_INSTS_LDS;SQ_WAIT_INST_ANY SQ_BUSY_CYCLES;FETCH_SIZE;WRITE_SIZE"
IFS=';' read -ra GROUPS_ARR <<< "${PMC_GROUPS:-$DEFAULT_GROUPS}"
for C in "${GROUPS_ARR[@]}"; do
  N=$(echo $C | tr ' ' '_')
  timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$N -- $CMD > $OUT/pmc_$N.log 2>&1 || echo "pass $N failed" >&2
done
python3 - <<PY
import csv, glob, json
acc = {}
for f in glob.glob("$OUT/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        targs = [x.strip() for x in k[k.find("<") + 1:k.find(">")].split(",")] if "<" in k else []
        if "rocclr" in k or "elementwise" in k or (len(targs) > 1 and targs[1] == "true"):   # (instrumented builds: STATS is the second argument)
            continue
        a = acc.setdefault(k[:60], {}).setdefault(r["Counter_Name"], [0.0, 0])
        a[0] += float(r["Counter_Value"]); a[1] += 1
out = {k: {c: v[0] / v[1] for c, v in d.items()} for k, d in acc.items()}
for k, d in out.items():
    if "TCC_HIT_sum" in d: d["l2_hit"] = d["TCC_HIT_sum"] / max(1.0, d["TCC_HIT_sum"] + d["TCC_MISS_sum"])
    if "SQ_ACTIVE_INST_VALU" in d and "GRBM_GUI_ACTIVE" in d: d["valu_busy"] = d["SQ_ACTIVE_INST_VALU"] * 4 / (d["GRBM_GUI_ACTIVE"] / 8 * 1024) if d["GRBM_GUI_ACTIVE"] else 0
print(json.dumps(out, indent=1))
PY
