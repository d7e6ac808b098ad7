#!/usr/bin/env python3
"""Summarise a tools/profile_bench.sh output directory: per-kernel time (kernel-trace stats) and PMC counters
for the render kernel, averaged per dispatch.  Usage: summarize_profile.py gpurun_out/prof_<tag> [> profiles/...]"""
import csv, glob, os, sys, collections
d = sys.argv[1]
def find(sub, pat):
    g = glob.glob(os.path.join(d, sub, "**", pat), recursive=True)
    return max(g, key=os.path.getmtime) if g else None   # a reused tag leaves older runs beside the new one: take the newest
ks = find("trace", "*kernel_stats.csv")
if ks:
    print("== kernel-trace --stats (", os.path.relpath(ks, d), ")")
    for r in csv.DictReader(open(ks)):
        name = r["Name"]
        if len(name) > 70: name = name[:67] + "..."
        print(f"  {name:70s} calls {r['Calls']:>4} total_ns {r['TotalDurationNs']:>12} avg_ns {float(r['AverageNs']):>14.0f} pct {r['Percentage']}")
kt = find("trace", "*kernel_trace.csv")
if kt:
    rows = [r for r in csv.DictReader(open(kt)) if "rt_render" in r["Kernel_Name"] or "rt_tier" in r["Kernel_Name"]]
    if rows:
        r = rows[-1]
        print("== render kernel launch:", {k: r[k] for k in r if k in ("Kernel_Name","VGPR_Count","Accum_VGPR_Count","SGPR_Count","LDS_Block_Size","Scratch_Size","Workgroup_Size_X","Grid_Size_X")})
        durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
        print("   durations ms:", [round(x, 3) for x in durs])
for sub in ("pmc_sq1", "pmc_sq2", "pmc_sq3", "pmc_fetch", "pmc_write"):
    f = find(sub, "*counter_collection.csv")
    if not f: continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "rt_render" not in r["Kernel_Name"] and "rt_tier" not in r["Kernel_Name"]: continue
        which = "tier kernel" if "rt_tier" in r["Kernel_Name"] else "main kernel"
        acc[(which, r["Counter_Name"])].append(float(r["Counter_Value"]))
    print(f"== {sub}: per-dispatch mean, by kernel (dispatches: {max((len(v) for v in acc.values()), default=0)})")
    for (which, k), v in sorted(acc.items()):
        print(f"  {which:12s} {k:28s} {sum(v)/len(v):>20.0f}   ({len(v)} dispatches)")
