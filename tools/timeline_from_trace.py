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
# the last frame: from its cost-prior kernel on, or -- without the prior -- from the third-last main-kernel launch (a split
# frame is three of them) and the buffer fills in front of it
starts = [i for i, r in enumerate(rows) if short(r[2]) == "prior"]
if starts:
    i0 = starts[-1]
else:
    mains = [i for i, r in enumerate(rows) if short(r[2]) == "main"]
    i0 = mains[-3] if len(mains) >= 3 else mains[0]
    while i0 > 0 and "fillBuffer" in rows[i0 - 1][2]: i0 -= 1
t0 = rows[i0][0]
print(f"{'kernel':12s} {'start ms':>9s} {'end ms':>9s} {'dur ms':>9s}  grid x wg   vgpr scratch lds")
for s, e, n, wg, grid, vg, sc, lds in rows[i0:]:
    print(f"{short(n):12s} {(s - t0) / 1e6:9.3f} {(e - t0) / 1e6:9.3f} {(e - s) / 1e6:9.3f}  {grid // max(wg, 1)} x {wg}  {vg} {sc} {lds}")
