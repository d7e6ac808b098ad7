#!/usr/bin/env python3
"""Per-launch timeline of the LAST frame in a rocprofv3 --kernel-trace CSV (of tools/one_frame.py): every dispatch from the
frame's first ranking / prior kernel on, with start and end relative to the frame's start.  Usage: timeline_from_trace.py <dir>"""
import csv, glob, os, sys
paths = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)
rows = []
for p in paths:
    with open(p) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r.get("Workgroup_Size", r.get("Workgroup_Size_X", 0)) or 0), int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0), r.get("VGPR_Count", "?"), r.get("Scratch_Size", r.get("Private_Segment_Size", "?")), r.get("LDS_Block_Size", "?")))
rows.sort()
def short(n):
    for key, s in (("rt_tier_kernel", "tier"), ("rt_render_staged_kernel", "main"), ("rt_render_pixel_kernel", "pixel"), ("rt_rank_tiles", "rank_tiles"), ("rt_collect_heavy", "collect"), ("rt_rank_heavy", "rank_heavy"), ("rt_prior", "prior"), ("rt_uninterleave", "uninterleave")):
        if key in n: return s
    return n[:40]
# frames are separated by gaps; a frame starts at a "prior" kernel (or the first main kernel after a long gap)
starts = [i for i, r in enumerate(rows) if short(r[2]) == "prior"]
if not starts:   # no cost prior: a frame starts at the first render kernel after a pause of the device (> 2 ms)
    starts = [i for i, r in enumerate(rows) if short(r[2]) in ("main", "tier") and (i == 0 or r[0] - max(x[1] for x in rows[:i]) > 2_000_000)]
i0 = starts[-1]
t0 = rows[i0][0]
print(f"{'kernel':12s} {'start ms':>9s} {'end ms':>9s} {'dur ms':>9s}  grid x wg   vgpr scratch lds")
for s, e, n, wg, grid, vg, sc, lds in rows[i0:]:
    print(f"{short(n):12s} {(s - t0) / 1e6:9.3f} {(e - t0) / 1e6:9.3f} {(e - s) / 1e6:9.3f}  {grid // max(wg, 1)} x {wg}  {vg} {sc} {lds}")
