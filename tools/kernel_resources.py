#!/usr/bin/env python3
"""Register / scratch / occupancy table of every kernel the render library ships.

Compiles each kernel translation unit under accelerated-ray-tracer_amd/csrc with the Makefile's flags plus
-Rpass-analysis=kernel-resource-usage (no GPU needed: hipcc cross-compiles gfx950) and prints one line per __global__
function.  Extra arguments go to hipcc (e.g. -DRT_LEAN_MIN_WAVES=5).  `--md` prints a markdown table (profiles/).
"""
import glob
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(root, "accelerated-ray-tracer_amd", "csrc")
args = [a for a in sys.argv[1:] if a != "--md"]
md = "--md" in sys.argv[1:]
units = sorted(glob.glob(os.path.join(csrc, "*.hip")))


def compile_unit(src):
    obj = f"/tmp/_rt_res_{os.path.basename(src)}.o"
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
           "-c", src, "-o", obj, "-Rpass-analysis=kernel-resource-usage"] + args
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stderr[-2000:])
        raise SystemExit(f"{src}: compile failed")
    return src, r.stderr


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return r.stdout.splitlines() if r.returncode == 0 else names


rows = []
with ThreadPoolExecutor(max_workers=min(8, len(units))) as ex:
    for src, out in ex.map(compile_unit, units):
        cur = None
        for line in out.splitlines():
            m = re.search(r"remark:\s+(.*?)\s+\[-Rpass", line)
            if not m:
                continue
            t = m.group(1)
            if t.startswith("Function Name:"):
                cur = {"unit": os.path.basename(src), "name": t.split(":", 1)[1].strip()}
                rows.append(cur)
            elif cur is not None and ":" in t:
                k, v = t.split(":", 1)
                cur[k.strip()] = v.strip()
# a remark block is emitted for device functions too (noinline helpers): keep the kernels
pretty = demangle([r["name"] for r in rows])
for r, p in zip(rows, pretty):
    p = re.sub(r"^void ", "", p)
    p = re.sub(r"\((anonymous namespace)::", "", p)
    p = re.sub(r"\(rt_scene_dev, rt_frame_params\)|\(rt_rank_params\)|\(.*\)$", "", p)
    r["short"] = p.replace("(anonymous namespace)::", "")
kernels = [r for r in rows if "kernel" in r["short"]]
if md:
    print("| translation unit | kernel | VGPR | AGPR | SGPR | scratch B/lane | waves/SIMD | LDS B/block (static) |")
    print("|---|---|---|---|---|---|---|---|")
for r in kernels:
    vals = (r["unit"], r["short"], r.get("VGPRs", "?"), r.get("AGPRs", "?"), r.get("TotalSGPRs", "?"),
            r.get("ScratchSize [bytes/lane]", "?"), r.get("Occupancy [waves/SIMD]", "?"), r.get("LDS Size [bytes/block]", "?"))
    if md:
        print("| " + " | ".join(f"`{v}`" if i == 1 else str(v) for i, v in enumerate(vals)) + " |")
    else:
        print(f"{vals[0]:28s} {vals[1]:64s} VGPR {vals[2]:>4} AGPR {vals[3]:>3} SGPR {vals[4]:>4} scratch {vals[5]:>5} occ {vals[6]:>2} LDS {vals[7]}")
