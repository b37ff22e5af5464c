"""Turn gpurun_out/prof_<tag>/ (tools/profile_gpu.sh) into profiles/<name>_kernel_stats.csv and
profiles/<name>_pmc.json.  Per-kernel counters are averaged over the un-instrumented launches
(kernel names without the <.., true> STATS instantiation)."""
import csv, glob, json, os, shutil, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, name = sys.argv[1], sys.argv[2]
src = os.path.join(REPO, "gpurun_out", "prof_" + tag)
dst = os.path.join(REPO, "profiles")
stats = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)[0]
shutil.copy(stats, os.path.join(dst, name + "_kernel_stats.csv"))
rows = list(csv.DictReader(open(stats)))
STEPS = 4   # un-instrumented steps per profiled run (1 warmup + 3 timed)
summary = {"round": 1, "name": name,
           "command": "tools/profile_gpu.sh %s (rocprofv3 --kernel-trace --stats, then one --pmc pass per counter group, of "
                      "`python3 bench.py --steps 3 --warmup 1 --no-cpu --schedule ...`)" % tag,
           "workload": "bunny.json 512x512 256spp depth 8 (BASELINE configs[1])",
           "kernel_avg_ms": {r["Name"]: round(float(r["AverageNs"]) * 1e-6, 4) for r in rows
                             if float(r["Percentage"]) > 0.05},
           "kernel_calls": {r["Name"]: int(r["Calls"]) for r in rows if float(r["Percentage"]) > 0.05},
           "counters_per_launch": {}, "launch_info": {}}
for f in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    acc = {}
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "rocclr" in k or ", true>" in k:
            continue
        a = acc.setdefault((k, r["Counter_Name"]), [0.0, 0])
        a[0] += float(r["Counter_Value"]); a[1] += 1
        summary["launch_info"][k] = {"grid": int(r["Grid_Size"]), "wg": int(r["Workgroup_Size"]), "lds": int(r["LDS_Block_Size"]),
                                     "scratch": int(r["Scratch_Size"]), "vgpr": int(r["VGPR_Count"]),
                                     "agpr": int(r["Accum_VGPR_Count"]), "sgpr": int(r["SGPR_Count"])}
    for (k, c), (s, n) in acc.items():
        summary["counters_per_launch"].setdefault(k, {})[c] = s / n
        summary["counters_per_launch"][k]["launches"] = n
tot = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0}
for k, c in summary["counters_per_launch"].items():
    for x in tot:
        tot[x] += c.get(x, 0.0) * c.get("launches", 0)
    if "TCC_HIT_sum" in c:
        c["l2_hit_rate"] = c["TCC_HIT_sum"] / max(1.0, c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
summary["hbm_bytes_per_step"] = int((2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024 / STEPS)
summary["hbm_bytes_note"] = ("sum over all kernels of a step of (2*FETCH_SIZE + WRITE_SIZE) KiB; the x2 is the gfx950 correction of "
                             "MI355X_MICROARCH.md (128-B requests tallied at 64 B), calibrated for wide coalesced reads - the node and "
                             "triangle gathers here are 16 B per lane and divergent, so the absolute is +-2x")
json.dump(summary, open(os.path.join(dst, name + "_pmc.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
