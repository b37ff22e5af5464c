#!/bin/bash
# Runs ON THE GPU BOX: vector-L1 (TCP) counters of configs[1]'s kernels, two --pmc passes; prints the lean quad kernel's per-launch averages.
# (The TA_* counters made rocprofv3 time out on this pool: left out.)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_mem; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for C in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" "TCP_TOTAL_ACCESSES_sum TCP_TCP_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-30)
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$N -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-others > $OUT/$N.log 2>&1 || echo "pass $N failed"
done
python3 - <<'PY'
import csv, glob, collections
for f in glob.glob('/root/repo/gpurun_out/pmc_mem/**/*counter_collection.csv', recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'path_trace_kernel<0, false, false, true, false>' in k:
            acc[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
    for k, v in acc.items():
        print({c: x / cnt[(k, c)] for c, x in v.items()})
PY
