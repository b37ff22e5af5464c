"""Per-kernel averages of the counters tools/pmc_quick.sh collected: python tools/pmc_kernels.py <tag> [kernel-name substring]"""
import csv, glob, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else "path_trace_kernel<false, false, false"
acc = {}
for f in glob.glob(os.path.join(REPO, "gpurun_out", "pmcq_" + tag, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if sub not in k:
            continue
        a = acc.setdefault(k, {}).setdefault(r["Counter_Name"], [0.0, 0])
        a[0] += float(r["Counter_Value"])
        a[1] += 1
for k, d in acc.items():
    o = {c: v[0] / v[1] for c, v in d.items()}
    print(k)
    print("  ", {a: round(b / 1e9, 4) for a, b in sorted(o.items())})
    if "SQ_ACTIVE_INST_VALU" in o and o.get("GRBM_GUI_ACTIVE"):
        print("   valu_busy %.3f  kernel cycles %.1f M" % (o["SQ_ACTIVE_INST_VALU"] * 4 / (o["GRBM_GUI_ACTIVE"] / 8 * 1024), o["GRBM_GUI_ACTIVE"] / 8e6))
