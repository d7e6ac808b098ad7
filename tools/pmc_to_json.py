#!/usr/bin/env python3
"""Turns a tools/profile_bench.sh output directory into the per-step counter record bench.py reads for its roofline
object.  Usage: pmc_to_json.py gpurun_out/prof_<tag> <workload key> [frames]  > profiles/pmc_<key>.json

Per step (= one frame = every dispatch of the main kernel and of the tier kernel for that frame: with the cost-aware schedule
two of the main kernel, each with a tier-1 launch beside it and a tail launch behind it) it sums, over those dispatches of each rocprofv3 pass, and divides by the number of frames in the pass (= main-kernel
dispatches / dispatches per step; counter passes serialise the kernels, so the two never share a counter window):
  SQ_INSTS_VALU, SQ_THREAD_CYCLES_VALU, SQ_ACTIVE_INST_VALU, SQ_WAVE_CYCLES, SQ_WAIT_ANY, SQ_WAIT_INST_ANY,
  SQ_ACTIVE_INST_ANY, SQ_INSTS_LDS, SQ_LDS_IDX_ACTIVE, SQ_LDS_BANK_CONFLICT, SQ_INSTS_SALU, GRBM_GUI_ACTIVE,
  FETCH_SIZE, WRITE_SIZE (KiB), and the kernel-trace durations.
HBM bytes per step = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: FETCH_SIZE is doubled as MI355X_MICROARCH.md (section HBM)
prescribes for gfx950 (uncalibrated for this access pattern -- the traffic is 0.01 % of the roof either way).  The same sum
over EVERY kernel of the frame, by kernel, is recorded as hbm_bytes_by_kernel: the accounting of the frame's HBM traffic.
"""
import csv, glob, json, os, sys, collections

d, key = sys.argv[1], sys.argv[2]
KERNEL = "rt_render_staged_kernel"          # the main kernel: its dispatches count the frames
RENDER_KERNELS = ("rt_render_staged_kernel", "rt_tier_kernel")


def short(name):
    for key in ("rt_tier_kernel", "rt_render_staged_kernel", "rt_rank_tiles", "rt_collect_heavy", "rt_rank_heavy", "rt_prior_kernel", "fillBuffer", "copyBuffer"):
        if key in name:
            return key
    return name.split("(")[0][:48]


def find(sub, pat):
    g = glob.glob(os.path.join(d, sub, "**", pat), recursive=True)
    return max(g, key=os.path.getmtime) if g else None   # a reused tag leaves older runs beside the new one: take the newest


def per_frame(sub):
    f = find(sub, "*counter_collection.csv")
    if not f:
        return {}, 0
    acc, disp = collections.defaultdict(float), set()
    for r in csv.DictReader(open(f)):
        by_kernel[short(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
        if not any(k in r["Kernel_Name"] for k in RENDER_KERNELS):
            continue
        acc[r["Counter_Name"]] += float(r["Counter_Value"])
        if KERNEL in r["Kernel_Name"]:
            disp.add(r["Dispatch_Id"])
    return acc, len(disp)


by_kernel = collections.defaultdict(lambda: collections.defaultdict(float))


out = {"workload": key, "source_dir": os.path.basename(d.rstrip("/")), "kernel": " + ".join(RENDER_KERNELS)}
kt = find("trace", "*kernel_trace.csv")
all_rows = list(csv.DictReader(open(kt)))
rows = [r for r in all_rows if KERNEL in r["Kernel_Name"]]
tier_rows = [r for r in all_rows if "rt_tier_kernel" in r["Kernel_Name"]]
durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
# dispatches per frame: a frame's last dispatch is its longest; frames = count of local maxima pattern -> use the argument or infer 3 / 1
# (a frame's last launch is by far its longest: launches per frame = launches / launches at least half as long as the longest)
long_ones = sum(1 for x in durs if x >= 0.5 * max(durs))
per = int(sys.argv[3]) if len(sys.argv) > 3 else (len(durs) // long_ones if long_ones and len(durs) % long_ones == 0 else 1)
frames = len(durs) // per
out["dispatches_per_step"] = per
out["frames_profiled"] = frames
out["kernel_ms_per_step_traced"] = round(sum(durs) / frames, 3)
out["dispatch_ms"] = [round(x, 3) for x in durs[-per:]]
out["vgpr_count"] = rows[-1].get("VGPR_Count")
if tier_rows:
    # the tier kernel's launches of the last frame, in start order: per ranked part one beside the main kernel (tier 1) and one
    # behind it (the tail launch) -- told apart by their grids
    t_first = int(rows[-per]["Start_Timestamp"]) - 1000000
    last = sorted((r for r in tier_rows if int(r["Start_Timestamp"]) >= t_first), key=lambda r: int(r["Start_Timestamp"]))
    out["tier_dispatch_ms"] = [round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, 3) for r in last]
    out["tier_dispatch_workgroups"] = [int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0) // max(int(r.get("Workgroup_Size", r.get("Workgroup_Size_X", 1)) or 1), 1) for r in last]
    out["tier_vgpr_count"] = tier_rows[-1].get("VGPR_Count")
counters = {}
for sub in ("pmc_sq1", "pmc_sq2", "pmc_sq3", "pmc_fetch", "pmc_write"):
    acc, nd = per_frame(sub)
    if not nd:
        continue
    fr = nd // per
    for k, v in acc.items():
        counters[k] = v / fr
out["per_step"] = {k: round(v, 1) for k, v in sorted(counters.items())}
if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
    out["hbm_bytes_per_step"] = int((2 * counters["FETCH_SIZE"] + counters["WRITE_SIZE"]) * 1024)
    # every kernel of the frame (ranking, prior, buffer fills and copies included), same correction
    out["hbm_bytes_by_kernel"] = {k: int((2 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024 / max(frames, 1))
                                  for k, v in sorted(by_kernel.items()) if v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0) > 0}
    out["hbm_bytes_per_step_all_kernels"] = sum(out["hbm_bytes_by_kernel"].values())
# which sources the counters belong to: recorded by tools/profile_bench.sh on the box that ran the passes
sha = os.path.join(d, "csrc_sha1.txt")
out["csrc_sha1"] = open(sha).read().strip() if os.path.exists(sha) else "unrecorded"
print(json.dumps(out, indent=1))
