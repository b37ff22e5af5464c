"""Turn gpurun_out/prof_<tag>/ (tools/profile_gpu.sh) into profiles/<name>_kernel_stats.csv and profiles/<name>.json.

    python tools/summarize_profile.py <tag> <name> [workload description]

Per-kernel counters are averaged over the un-instrumented launches (kernel names without a STATS = true template
argument are told apart by the caller's kernel tag; every kernel is listed).  The summary records the source stamp of
the tree it was collected from (goblin_amd/build.py source_stamp): bench.py quotes counters only when the stamp is the
running tree's.
"""
import csv
import glob
import json
import os
import shutil
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from goblin_amd import build

tag, name = sys.argv[1], sys.argv[2]
what = sys.argv[3] if len(sys.argv) > 3 else ""
src = os.path.join(REPO, "gpurun_out", "prof_" + tag)
dst = os.path.join(REPO, "profiles")
# (gpurun merges a call's files into gpurun_out/ beside older ones: always the newest of each kind)
stats = max(glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
shutil.copy(stats, os.path.join(dst, name + "_kernel_stats.csv"))
rows = list(csv.DictReader(open(stats)))
cmd = open(os.path.join(src, "command.txt")).read().strip().replace(REPO + "/", "") if os.path.exists(os.path.join(src, "command.txt")) else ""
# the stamp of the tree the passes ran on, written on the GPU box by tools/profile_gpu.sh; a summary made from a tree that has
# moved on since is refused rather than stamped with the wrong sources
stamp_file = os.path.join(src, "stamp.txt")
stamp = open(stamp_file).read().strip() if os.path.exists(stamp_file) else None
if stamp is None:
    sys.exit("no %s: re-collect with tools/profile_gpu.sh (it records the source stamp at collection time)" % stamp_file)
if stamp != build.source_stamp():
    print("warning: counters were collected from source stamp %s, this tree is %s -- the summary keeps the COLLECTED stamp, bench.py will not quote it" %
          (stamp, build.source_stamp()), file=sys.stderr)
import re
m_spp = re.search(r"--spp (\d+)", cmd)
summary = {"name": name, "source_stamp": stamp,
           # (a frame profiled at fewer samples per pixel than its BASELINE size: bench.py scales the extensive counters)
           "spp_profiled": int(m_spp.group(1)) if m_spp else None,
           "command": "tools/profile_gpu.sh %s %s  (rocprofv3 --kernel-trace --stats, then one --pmc pass per counter group)" % (tag, cmd),
           "workload": what,
           "kernel_avg_ms": {r["Name"]: round(float(r["AverageNs"]) * 1e-6, 4) for r in rows if float(r["Percentage"]) > 0.05},
           "kernel_calls": {r["Name"]: int(r["Calls"]) for r in rows if float(r["Percentage"]) > 0.05},
           "kernel_total_ms": {r["Name"]: round(float(r["TotalDurationNs"]) * 1e-6, 3) for r in rows if float(r["Percentage"]) > 0.05},
           "counters_per_launch": {}, "launch_info": {}}
newest = {}
for f in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    group = os.path.relpath(f, src).split(os.sep)[0]
    if group not in newest or os.path.getmtime(f) > os.path.getmtime(newest[group]):
        newest[group] = f
for f in newest.values():
    acc = {}
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "rocclr" in k or "elementwise" in k:
            continue
        a = acc.setdefault((k, r["Counter_Name"]), [0.0, 0])
        a[0] += float(r["Counter_Value"])
        a[1] += 1
        summary["launch_info"][k] = {"grid": int(r["Grid_Size"]), "wg": int(r["Workgroup_Size"]), "lds": int(r["LDS_Block_Size"]),
                                     "scratch": int(r["Scratch_Size"]), "vgpr": int(r["VGPR_Count"]),
                                     "agpr": int(r["Accum_VGPR_Count"]), "sgpr": int(r["SGPR_Count"])}
    for (k, c), (s, n) in acc.items():
        summary["counters_per_launch"].setdefault(k, {})[c] = s / n
        summary["counters_per_launch"][k]["launches"] = n
for k, c in summary["counters_per_launch"].items():
    if k in summary["kernel_avg_ms"]:
        c["kernel_avg_ms"] = summary["kernel_avg_ms"][k]
    if "TCC_HIT_sum" in c:
        c["l2_hit_rate"] = c["TCC_HIT_sum"] / max(1.0, c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        # MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 tallies 128-B read requests at 64 B (x2)
        c["hbm_bytes_per_launch"] = int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)
    if "SQ_ACTIVE_INST_VALU" in c and "GRBM_GUI_ACTIVE" in c and c["GRBM_GUI_ACTIVE"]:
        c["valu_busy_frac"] = c["SQ_ACTIVE_INST_VALU"] * 4 / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024)
    if "SQ_THREAD_CYCLES_VALU" in c and c.get("SQ_ACTIVE_INST_VALU"):
        # active-lane fraction of the VALU instructions: thread-cycles over 64 x the instructions' own (quad-)cycles
        c["lane_util_pmc"] = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])
    if "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"]:
        # MI355X_MICROARCH.md: WAIT_ANY (parked at s_waitcnt / barrier) + WAIT_INST_ANY (issue stall) + ACTIVE_INST_ANY ~ WAVE_CYCLES
        for key, out in (("SQ_WAIT_ANY", "wait_frac_pmc"), ("SQ_WAIT_INST_ANY", "issue_stall_frac_pmc"), ("SQ_ACTIVE_INST_ANY", "active_frac_pmc")):
            if key in c:
                c[out] = c[key] / c["SQ_WAVE_CYCLES"]
json.dump(summary, open(os.path.join(dst, name + ".json"), "w"), indent=1)
print(json.dumps({k: {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items()} for k, v in summary["counters_per_launch"].items()
                  if summary["kernel_avg_ms"].get(k, 0) > 0.2}, indent=1))
