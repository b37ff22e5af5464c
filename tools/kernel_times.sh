#!/bin/bash
# Runs ON THE GPU BOX: per-kernel average times of tools/render_once.py's renders (rocprofv3 --kernel-trace --stats).
#   tools/kernel_times.sh [scene res spp depth [exact]]
ROOT=$(pwd); export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/kt_prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_prof -- python3 $ROOT/tools/render_once.py "$@" 2>&1 | grep "kernel ms"
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/kt_prof/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if float(r["Percentage"]) > 0.1:
        print("%-110s calls %4s avg %.3f ms" % (r["Name"][:110], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
