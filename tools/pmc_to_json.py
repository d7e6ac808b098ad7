#!/usr/bin/env python3
"""Turns a tools/profile_bench.sh output directory into the per-step counter record bench.py reads for its roofline
object.  Usage: pmc_to_json.py gpurun_out/prof_<tag> <workload key> [frames]  > profiles/pmc_<key>.json

Per step (= one frame = every dispatch of the render kernel for that frame: three with the cost-aware schedule) it sums,
over the render-kernel dispatches of each rocprofv3 pass and divides by the number of frames in the pass:
  SQ_INSTS_VALU, SQ_THREAD_CYCLES_VALU, SQ_ACTIVE_INST_VALU, SQ_WAVE_CYCLES, SQ_WAIT_ANY, SQ_WAIT_INST_ANY,
  SQ_ACTIVE_INST_ANY, SQ_INSTS_LDS, SQ_LDS_IDX_ACTIVE, SQ_LDS_BANK_CONFLICT, SQ_INSTS_SALU, GRBM_GUI_ACTIVE,
  FETCH_SIZE, WRITE_SIZE (KiB), and the kernel-trace durations.
HBM bytes per step = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: FETCH_SIZE is doubled as MI355X_MICROARCH.md (section HBM)
prescribes for gfx950 (uncalibrated for this access pattern -- the traffic is 0.01 % of the roof either way).
"""
import csv, glob, json, os, sys, collections

d, key = sys.argv[1], sys.argv[2]
KERNEL = "rt_render_staged_kernel"


def find(sub, pat):
    g = glob.glob(os.path.join(d, sub, "**", pat), recursive=True)
    return max(g, key=os.path.getmtime) if g else None   # a reused tag leaves older runs beside the new one: take the newest


def per_frame(sub):
    f = find(sub, "*counter_collection.csv")
    if not f:
        return {}, 0
    acc, disp = collections.defaultdict(float), set()
    for r in csv.DictReader(open(f)):
        if KERNEL not in r["Kernel_Name"]:
            continue
        acc[r["Counter_Name"]] += float(r["Counter_Value"])
        disp.add(r["Dispatch_Id"])
    return acc, len(disp)


out = {"workload": key, "source_dir": os.path.basename(d.rstrip("/")), "kernel": KERNEL}
kt = find("trace", "*kernel_trace.csv")
rows = [r for r in csv.DictReader(open(kt)) if KERNEL in r["Kernel_Name"]]
durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
# dispatches per frame: a frame's last dispatch is its longest; frames = count of local maxima pattern -> use the argument or infer 3 / 1
per = int(sys.argv[3]) if len(sys.argv) > 3 else (3 if len(durs) % 3 == 0 and len(durs) >= 3 and durs[2] > 4 * durs[0] else 1)
frames = len(durs) // per
out["dispatches_per_step"] = per
out["frames_profiled"] = frames
out["kernel_ms_per_step_traced"] = round(sum(durs) / frames, 3)
out["dispatch_ms"] = [round(x, 3) for x in durs[-per:]]
out["vgpr_count"] = rows[-1].get("VGPR_Count")
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
# which sources the counters belong to: recorded by tools/profile_bench.sh on the box that ran the passes
sha = os.path.join(d, "csrc_sha1.txt")
out["csrc_sha1"] = open(sha).read().strip() if os.path.exists(sha) else "unrecorded"
print(json.dumps(out, indent=1))
