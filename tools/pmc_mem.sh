#!/bin/bash
# Runs ON THE GPU BOX: vector-L1 (TCP) counters of configs[1]'s kernels, two --pmc passes; prints the lean quad kernel's per-launch averages.
# The TA_* counters go two per pass.  Round 3 asked for four in one pass and recorded "rocprofv3 timed out": the pass's own log
# (gpurun_out/pmc_mem/TA_BUSY_avr_TA_ADDR_STALLED_BY.log) shows the profiler aborting 1.2 s after start, before any kernel ran --
# "rocprofiler_create_counter_config ... error code 38: Request exceeds the capabilities of the hardware to collect", raised inside
# the process's first HIP call (gbl_create's upload), then signal 6.  The texture-addresser block has two counter slots per pass
# on gfx950; nothing hung on the GPU.  A failed pass now prints the profiler's own reason.
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_mem; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for C in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" "TCP_TOTAL_ACCESSES_sum TCP_TCP_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
         "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-30)
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$N -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-others > $OUT/$N.log 2>&1 || { echo "pass $N failed:"; grep -m 3 -i "error code\|Could not\|not found\|invalid" $OUT/$N.log; }
done
python3 - <<'PY'
import csv, glob, collections
for f in glob.glob('/root/repo/gpurun_out/pmc_mem/**/*counter_collection.csv', recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'path_trace_kernel<0, false, false, true, false' in k or 'primary_kernel' in k:
            acc[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
    for k, v in acc.items():
        print(k.split('(')[0], {c: x / cnt[(k, c)] for c, x in v.items()})
PY
